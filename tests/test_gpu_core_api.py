"""Batched agents.cppmodule.core drop-ins (core_api.hip) vs the reference's own outputs (ref_uct.npz)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
from make_golden import random_dag  # noqa: E402

pytestmark = pytest.mark.gpu


def test_core_api_matches_reference_golden(golden_dir):
    import torch
    from tetris_mcts_amd import _lib
    from tetris_mcts_amd.store import norm_quantile_table, _p, _stream
    L = _lib.lib()
    gold = np.load(os.path.join(golden_dir, "ref_uct.npz"))
    rng = np.random.default_rng(20260925)
    dev = torch.device("cuda")
    nq = norm_quantile_table(1 << 16, dev)
    keep = []  # device temporaries must outlive the asynchronous launches that read them

    def d(a):
        t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        keep.append(t)
        return t
    # The golden generator interleaves rng draws with results (k depends on the trace), so replay case by case
    rng = np.random.default_rng(20260925)
    for ci in range(24):
        n_nodes, n_obs = 400, int(rng.integers(60, 300))
        child, n_to_o, score, visit, value, variance = random_dag(rng, n_nodes, n_obs, low_visits=(ci % 2 == 0))
        if ci == 3:
            child[1] = 0
            child[1][2] = 5
            visit[n_to_o[5]] = 1
        low = [1, 5][ci % 3 == 2]
        N = n_nodes
        padn = lambda a: np.concatenate([a, np.zeros(N - len(a), a.dtype)])  # noqa: E731
        t_child, t_n2o, t_score = d(child), d(n_to_o), d(score)
        t_visit, t_value, t_var = d(padn(visit)), d(padn(value)), d(padn(variance))
        st = np.zeros(32, np.uint32)
        # glibc srand(1) state through the oracle
        from oracle import binding as B
        o = np.zeros(36, np.int32)
        B.lib().orc_srand(B.ptr(o), 1)
        st[:31] = o[:31].view(np.uint32)
        st[31] = int(o[34]) | (int(o[35]) << 8)
        t_rng = d(st.view(np.int32))
        name = "c%d" % ci
        tcat, tlen, uniq = gold[name + "_tracecat"], gold[name + "_tracelen"], gold[name + "_uniq"]
        off = 0
        trace = torch.zeros(512, dtype=torch.int32, device=dev)
        tl = torch.zeros(1, dtype=torch.int32, device=dev)
        for rep in range(12):
            root = int(rng.integers(1, 60))
            t_root = d(np.array([root], np.int32))
            _lib.check(L.tm_core_select_trace_obs(1, N, _p(t_root), _p(t_child), _p(t_visit), _p(t_value), _p(t_var),
                                                  _p(t_score), _p(t_n2o), low, _p(t_rng), _p(nq), nq.numel(), _p(trace),
                                                  _p(tl), 512, _stream()), "select")
            n = int(tl.item())
            assert n == tlen[rep], (ci, rep, n, tlen[rep])
            assert np.array_equal(trace[:n].cpu().numpy(), tcat[off:off + n]), (ci, rep)
            off += n
            cn = torch.zeros(7, dtype=torch.int32, device=dev)
            co = torch.zeros(7, dtype=torch.int32, device=dev)
            cnt = torch.zeros(1, dtype=torch.int32, device=dev)
            _lib.check(L.tm_core_get_unique_child_obs(1, N, _p(t_root), _p(t_child), _p(t_score), _p(t_n2o), _p(cn),
                                                      _p(co), _p(cnt), _stream()), "unique")
            k = int(cnt.item())
            u = uniq[rep]
            assert u[15] == k and np.array_equal(u[1:1 + k], cn[:k].cpu().numpy()) and np.array_equal(u[8:8 + k], co[:k].cpu().numpy())
            mode = rep % 3
            if mode == 0:
                v0, r0 = float(rng.random() * 90), float(rng.random() * 500)
                _lib.check(L.tm_core_backup_trace_obs(1, N, _p(trace), _p(tl), 512, _p(t_visit), _p(t_value), _p(t_var),
                                                      _p(t_n2o), _p(t_score), _p(d(np.array([v0]))), _p(d(np.array([r0]))),
                                                      C.c_double(0.999), _stream()), "backup")
            else:
                leaf = d(np.array([int(trace[n - 1].item())], np.int32))
                _lib.check(L.tm_core_get_unique_child_obs(1, N, _p(leaf), _p(t_child), _p(t_score), _p(t_n2o), _p(cn),
                                                          _p(co), _p(cnt), _stream()), "unique")
                k = int(cnt.item())
                end = np.zeros(N, np.uint8)
                if mode == 2 and k:
                    end[int(cn[0].item())] = 1
                _v = np.zeros(7, np.float32)
                _r = np.zeros(7, np.float32)
                _v[:k] = (rng.random(k) * 60).astype(np.float32)
                _r[:k] = (rng.random(k) * 400).astype(np.float32)
                _lib.check(L.tm_core_backup_trace_obs_lp(1, N, _p(trace), _p(tl), 512, _p(t_visit), _p(t_value), _p(t_var),
                                                         _p(t_n2o), _p(t_score), _p(d(end)), _p(cn), _p(co), _p(cnt),
                                                         _p(d(_v)), _p(d(_r)), C.c_double(0.999), 0, 1, _stream()), "backup_lp")
        root = int(rng.integers(1, 60))
        mark = torch.zeros(N, dtype=torch.uint8, device=dev)
        queue = torch.zeros(N, dtype=torch.int32, device=dev)
        _lib.check(L.tm_core_get_all_childs(1, N, _p(d(np.array([root], np.int32))), _p(t_child), _p(mark), _p(queue),
                                            _stream()), "all_childs")
        assert np.array_equal(np.nonzero(mark.cpu().numpy())[0].astype(np.int32), gold[name + "_reach"])
        no = len(visit)
        assert t_visit.cpu().numpy()[:no].tobytes() == gold[name + "_visit1"].tobytes(), ci
        assert t_value.cpu().numpy()[:no].tobytes() == gold[name + "_value1"].tobytes(), ci
        assert t_var.cpu().numpy()[:no].tobytes() == gold[name + "_variance1"].tobytes(), ci


def test_core_api_mixture_backup(golden_dir):
    """backup_trace_obs_LP with mixture / non-averaged flags vs the reference's own outputs (ref_uct_mixture.npz)."""
    import torch
    from tetris_mcts_amd import _lib
    from tetris_mcts_amd.store import norm_quantile_table, _p, _stream
    from oracle import binding as B
    L = _lib.lib()
    gold = np.load(os.path.join(golden_dir, "ref_uct_mixture.npz"))
    dev = torch.device("cuda")
    nq = norm_quantile_table(1 << 16, dev)
    keep = []

    def d(a):
        t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        keep.append(t)
        return t
    rng = np.random.default_rng(77)
    N = 300
    for ci in range(8):
        child, n_to_o, score, visit, value, variance = random_dag(rng, N, 150, low_visits=False)
        padn = lambda a: np.concatenate([a, np.zeros(N - len(a), a.dtype)])  # noqa: E731
        t_child, t_n2o, t_score = d(child), d(n_to_o), d(score)
        t_visit, t_value, t_var = d(padn(visit)), d(padn(value)), d(padn(variance))
        o = np.zeros(36, np.int32)
        B.lib().orc_srand(B.ptr(o), 1)
        st = np.zeros(32, np.uint32)
        st[:31] = o[:31].view(np.uint32)
        st[31] = int(o[34]) | (int(o[35]) << 8)
        t_rng = d(st.view(np.int32))
        trace = torch.zeros(512, dtype=torch.int32, device=dev)
        tl = torch.zeros(1, dtype=torch.int32, device=dev)
        cn = torch.zeros(7, dtype=torch.int32, device=dev)
        co = torch.zeros(7, dtype=torch.int32, device=dev)
        cnt = torch.zeros(1, dtype=torch.int32, device=dev)
        for rep in range(10):
            root = int(rng.integers(1, 40))
            _lib.check(L.tm_core_select_trace_obs(1, N, _p(d(np.array([root], np.int32))), _p(t_child), _p(t_visit),
                                                  _p(t_value), _p(t_var), _p(t_score), _p(t_n2o), 1, _p(t_rng), _p(nq),
                                                  nq.numel(), _p(trace), _p(tl), 512, _stream()), "select")
            n = int(tl.item())
            if n > 1:
                n -= 1
                tl.fill_(n)
            leaf = d(np.array([int(trace[n - 1].item())], np.int32))
            _lib.check(L.tm_core_get_unique_child_obs(1, N, _p(leaf), _p(t_child), _p(t_score), _p(t_n2o), _p(cn), _p(co),
                                                      _p(cnt), _stream()), "unique")
            k = int(cnt.item())
            _v = np.zeros(7, np.float32)
            _r = np.zeros(7, np.float32)
            _v[:k] = (rng.random(k) * 60).astype(np.float32)
            _r[:k] = (rng.random(k) * 400).astype(np.float32)
            _lib.check(L.tm_core_backup_trace_obs_lp(1, N, _p(trace), _p(tl), 512, _p(t_visit), _p(t_value), _p(t_var),
                                                     _p(t_n2o), _p(t_score), _p(d(np.zeros(N, np.uint8))), _p(cn), _p(co),
                                                     _p(cnt), _p(d(_v)), _p(d(_r)), C.c_double(0.98), rep & 1,
                                                     1 if rep & 2 else 0, _stream()), "backup_lp")
        no = len(visit)
        assert t_visit.cpu().numpy()[:no].tobytes() == gold["m%d_visit1" % ci].tobytes(), ci
        assert t_value.cpu().numpy()[:no].tobytes() == gold["m%d_value1" % ci].tobytes(), ci
        assert t_var.cpu().numpy()[:no].tobytes() == gold["m%d_variance1" % ci].tobytes(), ci


def test_distribution_helpers_match_the_reference(golden_dir):
    """tm_dist_transform / tm_dist_mean_variance vs the reference's own core.h:387-449 outputs (ref_dist.npz), batched."""
    import torch
    from tetris_mcts_amd import _lib
    from tetris_mcts_amd.store import _p, _stream
    g = np.load(os.path.join(golden_dir, "ref_dist.npz"))
    L = _lib.lib()
    by_cfg = {}
    for i in range(int(g["n"])):
        key = (len(g["dist_%d" % i]), float(g["vmin_%d" % i]), float(g["vmax_%d" % i]), float(g["scale_%d" % i]))
        by_cfg.setdefault(key, []).append(i)
    for (bins, vmin, vmax, scale), idx in by_cfg.items():
        d = torch.from_numpy(np.stack([g["dist_%d" % i] for i in idx]).astype(np.float32)).cuda()
        sh = torch.from_numpy(np.array([float(g["shift_%d" % i]) for i in idx], np.float64)).cuda()
        out = torch.zeros_like(d)
        mv = torch.zeros(len(idx), 2, dtype=torch.float64, device="cuda")
        _lib.check(L.tm_dist_transform(len(idx), bins, _p(d), vmin, vmax, _p(sh), scale, _p(out), _stream()), "tm_dist_transform")
        _lib.check(L.tm_dist_mean_variance(len(idx), bins, _p(d), vmin, vmax, _p(mv), _stream()), "tm_dist_mean_variance")
        for j, i in enumerate(idx):
            assert out[j].cpu().numpy().tobytes() == g["out_%d" % i].astype(np.float32).tobytes(), i
            assert mv[j].cpu().numpy().tobytes() == g["mv_%d" % i].astype(np.float64).tobytes(), i


def test_distributional_numba_kernels_match_the_reference_functions(oracle, golden_dir):
    """tm_distpy_shift / tm_distpy_policy / tm_distpy_backup vs a pure-Python run of agents/core_distributional.py
    (ref_distpy.npz; `fastmath` numba code: float tolerance) and vs the oracle restatement (same arithmetic: bit for bit
    where no libm call is involved)."""
    import torch
    from tetris_mcts_amd import _lib
    from tetris_mcts_amd.store import _p, _stream
    from test_oracle_dist import ATOL, RTOL, distpy_cases
    L, OL = _lib.lib(), oracle.lib()
    cuda = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    # shift_distribution, batched by shape
    groups = {}
    for c in distpy_cases(golden_dir, "s", ("dist", "x", "vmax", "out")):
        groups.setdefault((len(c["dist"]), float(c["vmax"])), []).append(c)
    n = 0
    for (bins, vmax), cs in groups.items():
        d = cuda(np.stack([c["dist"] for c in cs]).astype(np.float32))
        x = cuda(np.array([float(c["x"]) for c in cs], np.float64))
        out = torch.zeros_like(d)
        _lib.check(L.tm_distpy_shift(len(cs), bins, _p(d), _p(x), 0.0, vmax, _p(out), _stream()), "tm_distpy_shift")
        got = out.cpu().numpy()
        for i, c in enumerate(cs):
            assert np.allclose(got[i], c["out"], rtol=RTOL, atol=ATOL), (bins, i)
            want = np.zeros(bins, np.float32)
            src = np.ascontiguousarray(c["dist"], np.float32)
            OL.orc_distpy_shift(oracle.ptr(src), bins, float(c["x"]), 0.0, vmax, oracle.ptr(want))
            assert got[i].tobytes() == want.tobytes(), (bins, i)
            n += 1
    assert n == 24
    # policy_dist: all 40 trees in one call
    cs = list(distpy_cases(golden_dir, "p", ("stats", "nodes", "cur", "out")))
    N = cs[0]["stats"].shape[0]
    cn = np.zeros((len(cs), 7), np.int32)
    for i, c in enumerate(cs):
        cn[i, :len(c["nodes"])] = c["nodes"]
    out = torch.zeros(len(cs), dtype=torch.int32, device="cuda")
    args = [cuda(cn), cuda(np.array([len(c["nodes"]) for c in cs], np.int32)),
            cuda(np.stack([c["stats"] for c in cs]).astype(np.float32)), cuda(np.array([float(c["cur"]) for c in cs], np.float64))]
    _lib.check(L.tm_distpy_policy(len(cs), N, *[_p(a) for a in args], _p(out), _stream()), "tm_distpy_policy")
    assert out.cpu().numpy().tolist() == [int(c["out"]) for c in cs]
    # backup_trace_distributional, batched by shape
    groups = {}
    for c in distpy_cases(golden_dir, "b", ("stats_in", "dist_in", "trace", "r", "leaf", "vmax", "stats_out", "dist_out")):
        groups.setdefault((c["dist_in"].shape, float(c["vmax"]), len(c["trace"])), []).append(c)
    n = 0
    for ((N, bins), vmax, tl), cs in groups.items():
        B = len(cs)
        ns, nd = cuda(np.stack([c["stats_in"] for c in cs])), cuda(np.stack([c["dist_in"] for c in cs]))
        tr, tlen = cuda(np.stack([c["trace"] for c in cs]).astype(np.int32)), cuda(np.full(B, tl, np.int32))
        r, leaf = cuda(np.array([float(c["r"]) for c in cs], np.float64)), cuda(np.stack([c["leaf"] for c in cs]).astype(np.float32))
        scratch = torch.zeros(B, bins, device="cuda")
        _lib.check(L.tm_distpy_backup(B, N, bins, _p(tr), _p(tlen), tl, _p(ns), _p(nd), _p(r), _p(leaf), 0.0, vmax, _p(scratch),
                                      _stream()), "tm_distpy_backup")
        gs, gd = ns.cpu().numpy(), nd.cpu().numpy()
        for i, c in enumerate(cs):
            assert np.allclose(gs[i], c["stats_out"], rtol=1e-5, atol=1e-4), (N, bins, i)
            assert np.allclose(gd[i], c["dist_out"], rtol=RTOL, atol=ATOL), (N, bins, i)
            os_, od = c["stats_in"].copy(), c["dist_in"].copy()
            t_, l_, sc = np.ascontiguousarray(c["trace"], np.int32), np.ascontiguousarray(c["leaf"], np.float32), np.zeros(bins, np.float32)
            OL.orc_distpy_backup(oracle.ptr(t_), tl, oracle.ptr(os_), oracle.ptr(od), bins, float(c["r"]), oracle.ptr(l_), 0.0, vmax,
                                 oracle.ptr(sc))
            assert gs[i].tobytes() == os_.tobytes() and gd[i].tobytes() == od.tobytes(), (N, bins, i)
            n += 1
    assert n == 12
