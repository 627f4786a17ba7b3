"""Training-side pieces on CPU: Yogi vs its defining recurrences, the Gaussian KL loss, and
train_data's early stopping / best-weights reload / output-bound rescale."""
import math

import numpy as np
import torch


def test_yogi_matches_defining_recurrences():
    from tetris_mcts_amd.train import Yogi
    torch.manual_seed(0)
    p = torch.nn.Parameter(torch.randn(5, 3, dtype=torch.float64))
    opt = Yogi([p], lr=1e-3, betas=(0.9, 0.999), eps=1e-3, weight_decay=1e-3)
    x = p.detach().clone().numpy()
    m = np.zeros_like(x)
    v = None
    for t in range(1, 8):
        g = torch.randn(5, 3, dtype=torch.float64)
        p.grad = g.clone()
        opt.step()
        gn = g.numpy()
        if v is None:
            v = gn * gn                     # v0 = g0^2 before weight decay (yogi.py:60)
        gn = gn + 1e-3 * x                   # coupled weight decay (yogi.py:70-71)
        m = 0.9 * m + 0.1 * gn
        g2 = gn * gn
        v = v - (1 - 0.999) * np.sign(v - g2) * g2
        denom = np.sqrt(v) / math.sqrt(1 - 0.999 ** t) + 1e-3
        x = x - (1e-3 / (1 - 0.9 ** t)) * m / denom
        assert np.allclose(p.detach().numpy(), x, rtol=1e-12, atol=1e-14)


def test_gaussian_kl_is_zero_at_the_target_and_positive_elsewhere():
    from tetris_mcts_amd.train import gaussian_kl
    mean, var = torch.tensor([[3.0]]), torch.tensor([[2.0]])
    assert abs(float(gaussian_kl(var, mean, var, mean))) < 1e-7
    assert float(gaussian_kl(var * 2, mean + 1, var, mean)) > 0


def test_train_data_learns_and_restores_best_weights(tmp_path, monkeypatch):
    from tetris_mcts_amd import model as M
    monkeypatch.setattr(M, "EXP_PATH", str(tmp_path) + "/")
    torch.manual_seed(1)
    mdl = M.Model_VV.__new__(M.Model_VV)
    mdl.device = torch.device("cpu")
    mdl.model = M.Net().eval()
    mdl.backend, mdl._flat, mdl._prepared, mdl._scratch, mdl.optimizer = "torch", None, None, None, None
    monkeypatch.setattr(M.Model_VV, "save", lambda self, filename=None, verbose=True: None)
    n = 300
    rng = np.random.default_rng(0)
    states = np.zeros((n, 1, 20, 10), np.float32)
    h = rng.integers(0, 12, n)
    for i in range(n):
        states[i, 0, 20 - h[i]:, :] = (rng.random((h[i], 10)) < 0.7)
    values = (2.0 + 3.0 * h[:, None]).astype(np.float32)          # value grows with the stack height
    variances = np.full((n, 1), 4.0, np.float32)
    weights = rng.integers(3, 50, (n, 1)).astype(np.float32)
    before = float(torch.mean((mdl.model(torch.from_numpy(states))[:, 0:1] - torch.from_numpy(values)) ** 2))
    res = mdl.train_data([states, values, variances, weights], iters_per_val=20, batch_size=64, max_iters=80, log=False)
    after = float(torch.mean((mdl.model(torch.from_numpy(states))[:, 0:1] - torch.from_numpy(values)) ** 2))
    assert res["iters"] >= 20 and after < before
    assert torch.allclose(mdl.model.out_ubound, torch.tensor([values.max(), variances.max()]))   # model_vv.py:228-229
    assert not mdl.model.training


# ---- pinned on the reference's own training code (tests/golden/ref_training.npz, make_golden.py gen_training) ----
def _gold():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_training.npz"))


def test_yogi_trajectories_match_the_reference_bit_for_bit():
    """model/yogi.py:39-90 run unmodified (legacy overload shims) on a fixed gradient sequence, fp64 and fp32."""
    from tetris_mcts_amd.train import Yogi
    g = _gold()
    for dt, name in ((torch.float64, "f64"), (torch.float32, "f32")):
        p = torch.nn.Parameter(torch.tensor(g["yogi_p0"], dtype=dt))
        opt = Yogi([p], lr=1e-3, eps=1e-3, weight_decay=1e-3)
        for i, gr in enumerate(g["yogi_grads"]):
            p.grad = torch.tensor(gr, dtype=dt)
            opt.step()
            assert p.detach().numpy().tobytes() == g["yogi_traj_" + name][i].tobytes(), (name, i)


def test_gaussian_kl_matches_the_reference_bit_for_bit():
    from tetris_mcts_amd.train import gaussian_kl
    g = _gold()
    vp, mp, var, mean = [torch.from_numpy(x.copy()) for x in g["ll_in"]]
    assert gaussian_kl(vp, mp, var, mean).numpy().tobytes() == g["ll_out"].tobytes()


def test_training_steps_match_the_reference_bit_for_bit():
    """Model.train (model.py:97-116) x 4 on the reference's Net: same losses, gradient norms and all 478342 parameters."""
    from tetris_mcts_amd import model as M
    from tetris_mcts_amd import train as T
    g = _gold()
    torch.set_num_threads(1)
    mdl = M.Model_VV.__new__(M.Model_VV)
    mdl.device, mdl.backend = torch.device("cpu"), "torch"
    mdl._flat = mdl._prepared = mdl._scratch = None
    mdl.optimizer = None
    mdl.model = M.Net()
    mdl.set_flat_params(g["tr_params0"])
    mdl.model.train()
    opt = mdl._optimizer()
    assert [len(grp["params"]) for grp in opt.state_dict()["param_groups"]] == [len(g["opt_group_params"])]   # 12, one group
    batch = [torch.from_numpy(g[k].copy()) for k in ("tr_states", "tr_values", "tr_variances", "tr_weights")]
    for i in range(4):
        opt.zero_grad()
        loss, _ = T.batch_loss(mdl.model, batch, weighted=True)
        loss.backward()
        gn = math.sqrt(sum(float(p.grad.norm(2)) ** 2 for p in mdl.model.parameters() if p.grad is not None))
        opt.step()
        # the first step is bit-identical; afterwards the reference's torch.jit.script'ed Net runs an optimised graph whose
        # gradients differ from eager mode in the last bit of a few elements (relative 2e-9 on the norm)
        if i == 0:
            assert float(loss.detach()) == g["tr_losses"][i] and abs(gn - g["tr_gnorms"][i]) <= 1e-12 * gn
        assert abs(float(loss.detach()) - g["tr_losses"][i]) <= 2e-7 * abs(g["tr_losses"][i]), (i, float(loss.detach()), g["tr_losses"][i])
        assert abs(gn - g["tr_gnorms"][i]) <= 1e-7 * gn
    mdl._flat = None
    diff = np.abs(mdl.flat_params().numpy().astype(np.float64) - g["tr_params4"].astype(np.float64))
    moved = np.abs(g["tr_params4"].astype(np.float64) - g["tr_params0"].astype(np.float64))
    assert diff.max() <= 1e-7 and moved.max() > 1e-3, (diff.max(), moved.max())      # 4 Yogi steps of ~1e-3 each
    mdl.model.eval()
    vm, vs = T.validation_loss(mdl.model, batch, weighted=True, chunk=20)
    assert abs(vm - g["tr_val"][0]) < 1e-6 and abs(vs - g["tr_val"][1]) < 1e-6


def test_checkpoint_round_trips_with_a_reference_layout_optimizer_state(tmp_path):
    """the reference builds Yogi over all 12 parameters (model_vv.py:132); a checkpoint written here must load
    into such an optimiser and vice versa, before and after the first training step."""
    from tetris_mcts_amd import model as M
    from tetris_mcts_amd.train import Yogi
    mdl = M.Model_VV.__new__(M.Model_VV)
    mdl.device, mdl.backend = torch.device("cpu"), "torch"
    mdl._flat = mdl._prepared = mdl._scratch = None
    mdl.optimizer = None
    mdl.model = M.Net()
    path = str(tmp_path / "ck")
    mdl.save(path, verbose=False)                       # no optimiser existed yet: a real state dict is written anyway
    ck = torch.load(path)
    ref_style = Yogi(M.Net().parameters(), lr=1e-3, eps=1e-3, weight_decay=1e-3)
    ref_style.load_state_dict(ck["optimizer_state_dict"])          # what the reference's Model.load does (model.py:166-170)
    assert len(ck["optimizer_state_dict"]["param_groups"][0]["params"]) == 12
    torch.save({"model_state_dict": mdl.model.state_dict(), "optimizer_state_dict": ref_style.state_dict()}, path)
    mdl2 = M.Model_VV.__new__(M.Model_VV)
    mdl2.device, mdl2.backend = torch.device("cpu"), "torch"
    mdl2._flat = mdl2._prepared = mdl2._scratch = None
    mdl2.optimizer = None
    mdl2.model = M.Net()
    mdl2.load(path)
    assert mdl2.optimizer is not None


def test_distributional_head_fit_and_loss():
    """Model_Dist.loss = KL(target || softmax) with 0 log 0 = 0 (the reference's expression, model_distributional.py:84-98, is NaN
    at empty target atoms), weighted per sample; train_data (the generic train loop with this loss) lowers it."""
    import torch
    from tetris_mcts_amd.model_distributional import Model_Dist
    torch.manual_seed(3)
    m = Model_Dist(atoms=50, device="cpu", seed=1, backend="torch")
    n = 300
    x = torch.zeros(n, 1, 22, 10)
    x[:, 0, 12:, :] = (torch.rand(n, 10, 10) < 0.5).float()
    t = torch.zeros(n, 50)
    t[torch.arange(n), (x.sum((1, 2, 3)) / 4).long().clamp(max=49)] = 0.75      # a target with empty atoms
    t[torch.arange(n), ((x.sum((1, 2, 3)) / 4).long() + 1).clamp(max=49)] += 0.25
    w = torch.rand(n, 1) + 0.5
    mean, std = m.loss(x, t, w)
    ref = (w * (torch.where(t > 0, t * t.clamp(min=1e-30).log(), torch.zeros_like(t)) - t * m.model.log_prob(x))).sum(1)
    assert torch.isfinite(mean) and torch.allclose(mean, ref.mean(), rtol=1e-6)
    before = float(m.loss(x, t)[0])
    m.train_data([x, t, w], batch_size=64, iters_per_val=50, max_iters=150, log=False)
    assert float(m.loss(x, t)[0]) < 0.8 * before


def test_distributional_loss_is_the_reference_expression_where_that_is_finite():
    """model/model_distributional.py:84-98: loss = -value * (log p - log value), summed over the atoms, mean / std over the
    batch - identical numbers for targets without empty atoms (where the reference's expression is finite)."""
    import torch
    from tetris_mcts_amd.model_distributional import Model_Dist
    torch.manual_seed(0)
    m = Model_Dist(atoms=50, device="cpu", seed=2, backend="torch")
    x = (torch.rand(33, 1, 22, 10) < 0.3).float()
    t = torch.softmax(torch.randn(33, 50), 1)
    lp = m.model.log_prob(x)
    ref = (-t * (lp - t.log())).sum(dim=1)
    std, mean = torch.std_mean(ref)
    got_mean, got_std = m.loss(x, t)
    assert torch.allclose(got_mean, mean, rtol=1e-5, atol=1e-7) and torch.allclose(got_std, std, rtol=1e-4, atol=1e-7)
