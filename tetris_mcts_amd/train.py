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
    decay added to the gradient, bias-corrected step  p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps).

    Parameters on the GPU: the step is ONE HIP kernel over flat buffers (csrc/yogi.hip `tm_yogi_step`: the same operations per
    element in the same order) - the parameters, their gradients and the two moments become views of four flat tensors at the
    first step (`flatten`), the step count lives on the device, and an iteration of the fit is a fixed sequence of launches
    that train_data replays from a HIP graph.  The state dict keeps the reference's layout (per parameter: step, exp_avg,
    exp_avg_sq), so checkpoints load on either side.  Parameters on the CPU: the reference's own sequence of tensor operations."""

    def __init__(self, params, lr=1e-2, betas=(0.9, 0.999), eps=1e-3, weight_decay=0.0, fused=None):
        if lr <= 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1):
            raise ValueError("invalid Yogi hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._fused_wanted = fused
        self._flat = None

    # ---- the fused form ----
    def fused(self):
        """whether step() is the one-kernel form: every parameter on one GPU, float32, a single group (or asked for / refused
        at construction)"""
        if self._fused_wanted is False:
            return False
        ps = [p for g in self.param_groups for p in g["params"] if p.requires_grad]
        ok = (len(self.param_groups) == 1 and len(ps) > 0 and all(p.is_cuda and p.dtype == torch.float32 for p in ps)
              and len({p.device for p in ps}) == 1)
        if self._fused_wanted and not ok:
            raise ValueError("Yogi(fused=True) needs float32 parameters on one GPU in a single group")
        return ok

    def flatten(self):
        """(idempotent) the parameters that take gradients, their gradients and moments as views of flat buffers"""
        if self._flat is not None:
            return self._flat
        ps = [p for p in self.param_groups[0]["params"] if p.requires_grad]
        dev, n = ps[0].device, sum(p.numel() for p in ps)
        F = dict(params=ps, n=n, p=torch.empty(n, device=dev), g=torch.zeros(n, device=dev), m=torch.zeros(n, device=dev),
                 v=torch.zeros(n, device=dev), state=torch.zeros(8, dtype=torch.float64, device=dev))
        off, steps = 0, set()
        with torch.no_grad():
            for p in ps:
                k = p.numel()
                sl = slice(off, off + k)
                F["p"][sl].copy_(p.reshape(-1))
                p.data = F["p"][sl].view_as(p)
                if p.grad is not None:
                    F["g"][sl].copy_(p.grad.reshape(-1))
                p.grad = F["g"][sl].view_as(p)
                st = self.state[p]
                if st:       # moments loaded from a checkpoint (or steps taken by the per-tensor form)
                    F["m"][sl].copy_(st["exp_avg"].reshape(-1))
                    F["v"][sl].copy_(st["exp_avg_sq"].reshape(-1))
                    steps.add(int(st["step"]))
                st["exp_avg"], st["exp_avg_sq"] = F["m"][sl].view_as(p), F["v"][sl].view_as(p)
                st.setdefault("step", 0)
                off += k
        if len(steps) > 1:
            raise ValueError("Yogi: the parameters' step counts differ (%s)" % sorted(steps))
        t = steps.pop() if steps else 0
        b1, b2 = self.param_groups[0]["betas"]
        lr = self.param_groups[0]["lr"]
        if t > 0:
            F["state"].copy_(torch.tensor([t, b1 ** t, b2 ** t, lr / (1 - b1 ** t), math.sqrt(1 - b2 ** t), 0, 0, 0], dtype=torch.float64))
        F["t"] = t
        self._flat = F
        return F

    def flat_grad(self):
        return self.flatten()["g"] if self.fused() else None

    def zero_grad(self, set_to_none=True):
        if self._flat is not None:
            self._flat["g"].zero_()      # (the gradients stay the views they are: autograd accumulates into them in place)
            return
        super().zero_grad(set_to_none=set_to_none)

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        if self._flat is not None:       # the loaded moments are tensors of their own: back into the flat buffers
            ps = self._flat["params"]
            with torch.no_grad():
                for p in ps:
                    p.data = p.data.clone()
                    p.grad = None
            self._flat = None
            self.flatten()

    def state_dict(self):
        if self._flat is not None:
            for p in self._flat["params"]:
                self.state[p]["step"] = self._flat["t"]
        return super().state_dict()

    def _fused_step(self):
        from . import _lib
        F = self.flatten()
        g = self.param_groups[0]
        _lib.check(_lib.lib().tm_yogi_step(F["p"].data_ptr(), F["g"].data_ptr(), F["m"].data_ptr(), F["v"].data_ptr(),
                                           F["state"].data_ptr(), F["n"], float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]),
                                           float(g["eps"]), float(g["weight_decay"]), torch.cuda.current_stream(F["p"].device).cuda_stream),
                   "tm_yogi_step")
        F["t"] += 1

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self.fused():
            self._fused_step()
            return loss
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
    """Weighted combination over chunks (model.py:49-84): chunk weight = sum of sample weights (or the count).  The chunks'
    (weight, mean, std) stay on the device until the last one is in: ONE host synchronisation per validation (r05: three per
    chunk, ~150 per validation of a 50 000-tuple set)."""
    loss_fn = loss_fn or batch_loss
    rows = []
    for c in range(0, data[0].shape[0], chunk):
        b = [d[c:c + chunk] for d in data]
        mean, std = loss_fn(net, b, weighted)
        w = b[-1].sum() if weighted else torch.full((), float(b[0].shape[0]), device=mean.device)
        rows.append(torch.stack([w.to(torch.float64), mean.to(torch.float64), std.to(torch.float64)]))
    tot_w, acc, acc2 = 0.0, 0.0, 0.0
    for w, mean, std in torch.stack(rows).cpu().tolist():      # (the reference's arithmetic, in Python doubles as there)
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
    # the one-kernel optimiser step (Yogi on the GPU): parameters and gradients are views of flat buffers from here on
    flat_g = optimizer.flat_grad() if hasattr(optimizer, "flat_grad") else None

    def one_iteration():
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
            if flat_g is not None:
                tdist.all_reduce(flat_g, group=group)
                flat_g.div_(world)
            else:
                grads = [p.grad for p in net.parameters() if p.grad is not None]
                flat = torch.cat([g.reshape(-1) for g in grads])
                tdist.all_reduce(flat, group=group)
                flat /= world
                off = 0
                for g in grads:
                    g.copy_(flat[off:off + g.numel()].view_as(g))
                    off += g.numel()
        # 2-norm over all parameter gradients (model.py:87-95), reported in the log line the dashboards parse
        if flat_g is not None:
            g_norm_avg.add_(torch.linalg.vector_norm(flat_g))
        else:
            g_norm_avg.add_(torch.sqrt(sum((p.grad.detach() ** 2).sum() for p in net.parameters() if p.grad is not None)))
        if grad_clip > 0:
            torch.nn.utils.clip_grad_norm_(net.parameters(), grad_clip)
        optimizer.step()
        loss_avg.add_(loss.detach())

    # One iteration is a fixed sequence of launches (index draw, gather, forward, backward, the one-kernel step): after three
    # of them it is captured in a HIP graph and replayed - the fit of a 478 342-parameter net on batches of 1 024 was bound by
    # its launches, not by its arithmetic (r05: 2.3 ms an iteration for 0.1 ms of matrix work).  Not with more than one rank
    # (the all-reduce stays outside), a sampling stream of the caller's, or gradient clipping.
    import os
    graph, n_warm = None, 3
    want_graph = (flat_g is not None and world == 1 and flat_g.is_cuda and generator is None and grad_clip <= 0 and not oversampling
                  and sample_replacement and os.environ.get("TM_TRAIN_GRAPH", "1") != "0" and max_iters > n_warm)
    it = 0
    while it < max_iters:
        if want_graph and graph is None and it == n_warm:
            try:
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                t_before = optimizer._flat["t"]
                with torch.cuda.graph(graph):
                    one_iteration()
                optimizer._flat["t"] = t_before          # (captured, not run)
            except Exception as e:                        # noqa: BLE001 - whatever the runtime refuses: the eager loop is the same fit
                graph, want_graph = None, False
                torch.cuda.synchronize()
                if log:
                    print("train_data: no graph replay (%s: %s)" % (type(e).__name__, str(e).splitlines()[0][:120]), file=stderr, flush=True)
        if graph is not None:
            graph.replay()
            optimizer._flat["t"] += 1
        else:
            one_iteration()
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
        it += 1
    replayed = graph is not None
    del graph
    if early_stopping and load and best < float("inf"):
        load()
    elif save:
        save()
    net.eval()
    return dict(iters=iters_done, best_validation=best, graph_replay=replayed)
