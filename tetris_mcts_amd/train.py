"""Online TD training of the value net — the reference's learning half, as plain PyTorch on
the GPU (plumbing, not a hot kernel):

  * Yogi optimiser                      model/yogi.py:39-90   (lr 1e-3, eps 1e-3, weight decay 1e-3: model_vv.py:132)
  * Gaussian KL / likelihood loss       model/model_vv.py:94-101,136-150 (variance clipped at 0.1, optional weights)
  * Model.train_data                    model/model.py:176-249 (weights / mean, last 10 % validation, random minibatches
                                        with replacement, validation every `iters_per_val`, early stopping with patience,
                                        best-checkpoint reload)
  * Model_VV.train_data                 model/model_vv.py:227-231 (output upper bounds = data maxima)
  * checkpoint format                   model/model.py:143-174 ({'model_state_dict', 'optimizer_state_dict'})
"""
import math
from sys import stderr

import torch
from torch.optim.optimizer import Optimizer

variance_bound = 1e-1


class Yogi(Optimizer):
    """Yogi (Zaheer et al. 2018) with the reference's conventions: v0 = g0^2 (before weight decay), coupled weight
    decay added to the gradient, bias-corrected step  p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)."""

    def __init__(self, params, lr=1e-2, betas=(0.9, 0.999), eps=1e-3, weight_decay=0.0):
        if lr <= 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1):
            raise ValueError("invalid Yogi hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                g = p.grad
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = g * g
                st["step"] += 1
                t = st["step"]
                # the reference's own sequence of tensor primitives (model/yogi.py:69-88), so parameter trajectories are
                # bit-identical to it (tests/golden/ref_training.npz)
                if group["weight_decay"] != 0:
                    g = g.add(p, alpha=group["weight_decay"])
                m, v = st["exp_avg"], st["exp_avg_sq"]
                m.mul_(b1).add_(g, alpha=1 - b1)
                g2 = g.mul(g)
                v.addcmul_(torch.sign(v - g2), g2, value=-(1 - b2))
                denom = (v.sqrt() / math.sqrt(1 - b2 ** t)).add_(group["eps"])
                p.addcdiv_(m, denom, value=-(group["lr"] / (1 - b1 ** t)))
        return loss


def gaussian_kl(var_pred, mean_pred, var, mean):
    """KL( N(mean,var) || N(mean_pred,var_pred) ) up to the factor 1/2 the reference also drops."""
    return var_pred.log() + ((mean - mean_pred) ** 2 + var) / var_pred - var.log() - 1.0


def batch_loss(net, batch, weighted, variance_clip=variance_bound):
    state, value, variance, weight = batch
    variance = variance.clamp(min=variance_clip)
    out = net(state)
    v, var = out[:, 0:1], out[:, 1:2]
    per = gaussian_kl(var, v, variance, value)
    if weighted:
        per = weight * per
    std, mean = torch.std_mean(per, unbiased=False)      # model_vv.py:146-149 (one fused reduction: same bits as the reference)
    return mean, std


@torch.no_grad()
def validation_loss(net, data, weighted, chunk=1024, loss_fn=None):
    """Weighted combination over chunks (model.py:49-84): chunk weight = sum of sample weights (or the count)."""
    tot_w, acc, acc2 = 0.0, 0.0, 0.0
    loss_fn = loss_fn or batch_loss
    for c in range(0, data[0].shape[0], chunk):
        b = [d[c:c + chunk] for d in data]
        mean, std = loss_fn(net, b, weighted)
        w = float(b[-1].sum()) if weighted else float(b[0].shape[0])
        mean, std = float(mean), float(std)
        if math.isnan(std):
            std = 0.0
        tot_w += w
        acc += w * mean
        acc2 += w * (std * std + mean * mean)
    mean = acc / tot_w
    return mean, math.sqrt(max(acc2 / tot_w - mean * mean, 0.0))


def check_data_parallel_batch(batch_size, world):
    """Equal shards only: the average of the ranks' shard means is then the batch mean (with unequal shards it is not, and
    a rank with an empty shard would feed NaN into the all-reduce)."""
    if world > 1 and batch_size % world:
        raise ValueError("data-parallel training needs batch_size (%d) to be a multiple of the number of ranks (%d)"
                         % (batch_size, world))


def train_data(net, optimizer, data, batch_size=128, iters_per_val=500, validation_fraction=0.1,
               sample_replacement=True, oversampling=False, weighted=True, early_stopping=True, early_stopping_patience=10,
               early_stopping_threshold=1.0, shuffle=False, max_iters=100000, grad_clip=0.0, save=None, load=None,
               generator=None, log=True, data_parallel=True, group=None, loss_fn=None):
    """data = [states f32 [n,1,20,10], values [n,1], variances [n,1], weights [n,1]] (device tensors); with `loss_fn`
    (net, batch, weighted) -> (mean, std) any list of arrays whose LAST one holds the sample weights (Model.train_data is
    generic in the reference too, model/model.py:176-249: the model class supplies `_loss`).
    save() / load() persist and restore the best weights (the reference goes through its checkpoint file).

    With torch.distributed initialised and more than one rank (every rank holding the same data and the same weights, as
    after dist.all_gather_tuples), `data_parallel` splits each batch: every rank draws the SAME batch_size indices (one
    sampling stream shared by the ranks: `generator`, or one seeded by a number rank 0 broadcasts) and takes every
    world-th of them, the flattened gradients (1.9 MB) are averaged with one all-reduce per iteration, and every rank takes
    the same optimizer step - the replicas stay bit-identical and one iteration sees batch_size distinct draws, as in a
    single process.  Validation runs on every rank (same numbers, same stopping)."""
    import torch.distributed as tdist
    world = tdist.get_world_size(group) if (data_parallel and tdist.is_available() and tdist.is_initialized()) else 1
    my_rank = tdist.get_rank(group) if world > 1 else 0
    check_data_parallel_batch(batch_size, world)
    if world > 1 and generator is None:
        seed = torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).to(data[0].device)
        tdist.broadcast(seed, src=tdist.get_global_rank(group, 0) if group is not None else 0, group=group)
        generator = torch.Generator(device=data[0].device).manual_seed(int(seed.item()))
    n = data[0].shape[0]
    n_val = int(n * validation_fraction)
    data = list(data)
    data[-1] = data[-1] / data[-1].mean()
    loss_fn = loss_fn or batch_loss
    if shuffle:
        perm = torch.randperm(n, device=data[0].device, generator=generator)
        data = [d[perm] for d in data]
    train = [d[:n - n_val] for d in data] if n_val else data
    val = [d[n - n_val:] for d in data] if n_val else None
    if log:
        print("Training data size: {}    Validation data size: {}".format(n - n_val, n_val), file=stderr, flush=True)
    fails, best = 0, float("inf")
    loss_avg = torch.zeros((), device=data[0].device)      # running sums stay on the device: no sync per iteration
    g_norm_avg = torch.zeros((), device=data[0].device)
    iters_done = 0
    net.train()
    for it in range(max_iters):
        if oversampling:    # model.py:194-195: draw proportionally to the visit weights
            idx = torch.multinomial(train[-1].reshape(-1), batch_size, replacement=sample_replacement, generator=generator)
        elif sample_replacement:
            idx = torch.randint(0, n - n_val, (batch_size,), device=data[0].device, generator=generator)
        else:
            idx = torch.randperm(n - n_val, device=data[0].device, generator=generator)[:batch_size]
        if world > 1:
            idx = idx[my_rank::world]
        optimizer.zero_grad(set_to_none=True)
        loss, _ = loss_fn(net, [d[idx] for d in train], weighted)
        loss.backward()
        if world > 1:
            grads = [p.grad for p in net.parameters() if p.grad is not None]
            flat = torch.cat([g.reshape(-1) for g in grads])
            tdist.all_reduce(flat, group=group)
            flat /= world
            off = 0
            for g in grads:
                g.copy_(flat[off:off + g.numel()].view_as(g))
                off += g.numel()
        # 2-norm over all parameter gradients (model.py:87-95), reported in the log line the dashboards parse
        g_norm_avg += torch.sqrt(sum((p.grad.detach() ** 2).sum() for p in net.parameters() if p.grad is not None))
        if grad_clip > 0:
            torch.nn.utils.clip_grad_norm_(net.parameters(), grad_clip)
        optimizer.step()
        loss_avg += loss.detach()
        iters_done = it + 1
        if (it + 1) % iters_per_val == 0 and val is not None:
            net.eval()
            vmean, vstd = validation_loss(net, val, weighted, loss_fn=loss_fn)
            net.train()
            vstd /= max(n_val, 1) ** 0.5
            mark = ""
            if early_stopping:
                if vmean - best < vstd * early_stopping_threshold:
                    fails = 0
                    if vmean < best:
                        mark = "*"
                        best = vmean
                        if save:
                            save()
                else:
                    fails += 1
            if world > 1:      # the logged training loss is the mean over ranks (the gradient norm already is global)
                tdist.all_reduce(loss_avg, group=group)
                loss_avg /= world
            if log:
                print("Iteration:{:7d}  training loss:{:6.4f}  validation loss:{:6.4f}±{:6.4f}  gradient norm:{:6.3f}    {}"
                      .format(it + 1, float(loss_avg) / iters_per_val, vmean, vstd, float(g_norm_avg) / iters_per_val, mark),
                      file=stderr, flush=True)
            loss_avg.zero_()
            g_norm_avg.zero_()
            if early_stopping and fails >= early_stopping_patience:
                break
    if early_stopping and load and best < float("inf"):
        load()
    elif save:
        save()
    net.eval()
    return dict(iters=iters_done, best_validation=best)
