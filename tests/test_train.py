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
