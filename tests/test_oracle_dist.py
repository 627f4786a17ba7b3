"""The distribution helpers of the reference (agents/cppmodule/core.h:387-449, BASELINE configs[4]'s pinnable part):
oracle/dist_oracle.c against the reference's own outputs (tests/golden/ref_dist.npz, make_golden.py gen_dist)."""
import os

import numpy as np


def _cases(golden_dir):
    g = np.load(os.path.join(golden_dir, "ref_dist.npz"))
    for i in range(int(g["n"])):
        yield {k: g["%s_%d" % (k, i)] for k in ("dist", "vmin", "vmax", "shift", "scale", "out", "mean", "mv")}


def test_dist_oracle_matches_reference_bit_for_bit(oracle, golden_dir):
    L = oracle.lib()
    n = 0
    for c in _cases(golden_dir):
        d = np.ascontiguousarray(c["dist"], np.float32)
        bins = len(d)
        out = np.zeros(bins, np.float32)
        L.orc_transform_distribution(oracle.ptr(d), bins, float(c["vmin"]), float(c["vmax"]), float(c["shift"]), float(c["scale"]),
                                     oracle.ptr(out))
        assert out.tobytes() == c["out"].astype(np.float32).tobytes(), n
        assert L.orc_mean_dist(oracle.ptr(d), bins, float(c["vmin"]), float(c["vmax"])) == float(c["mean"])
        mv = np.zeros(2, np.float64)
        L.orc_mean_variance_dist(oracle.ptr(d), bins, float(c["vmin"]), float(c["vmax"]), oracle.ptr(mv))
        assert mv.tobytes() == c["mv"].astype(np.float64).tobytes(), n
        n += 1
    assert n == 48


def test_transform_distribution_drops_the_mass_beyond_the_top(oracle):
    """Where the reference writes result[bins] (one past its vector) the restatement drops that mass."""
    L = oracle.lib()
    d = np.full(50, 0.02, np.float32)
    out = np.zeros(50, np.float32)
    L.orc_transform_distribution(oracle.ptr(d), 50, 0.0, 100.0, 3.0, 0.99, oracle.ptr(out))
    assert 0.9 < out.sum() < 1.0 and out[0] == 0.0 and out[1] > 0


def test_distributional_net_matches_the_reference(golden_dir):
    """tetris_mcts_amd.model_distributional.Net loads the reference Net's state_dict as is and reproduces its outputs
    (CPU, same torch ops: bit for bit)."""
    import torch
    from tetris_mcts_amd.model_distributional import Net
    g = np.load(os.path.join(golden_dir, "ref_distnet.npz"))
    net = Net(atoms=50).eval()
    sd = {k.replace("__", "."): torch.from_numpy(g[k]) for k in g.files if k not in ("x", "y", "lp")}
    assert sorted(sd) == sorted(net.state_dict())
    net.load_state_dict(sd)
    torch.set_num_threads(1)
    with torch.no_grad():
        y = net(torch.from_numpy(g["x"])).numpy()
        lp = net.log_prob(torch.from_numpy(g["x"])).numpy()
    assert y.tobytes() == g["y"].tobytes() and lp.tobytes() == g["lp"].tobytes()
    assert np.allclose(y.sum(1), 1.0, atol=1e-6)


def test_distnet_oracle_is_pinned_on_the_reference_net(oracle, golden_dir):
    """oracle/distnet_oracle.c (one fp32 fma chain per pre-activation, the contract of csrc/distnet.hip) against the
    reference's own Net run on CPU (model/model_distributional.py:18-57 -> tests/golden/ref_distnet.npz): 1e-6 relative."""
    g = np.load(os.path.join(golden_dir, "ref_distnet.npz"))
    keys = ["seq__conv1__weight", "seq__conv1__bias", "seq__conv2__weight", "seq__conv2__bias", "seq__fc1__weight",
            "seq__fc1__bias", "seq__fc_v__weight", "seq__fc_v__bias"]
    P = np.concatenate([g[k].ravel() for k in keys]).astype(np.float32)
    assert P.size == 279232 + 129 * 50
    x = g["x"]
    assert (x[:, 0, :2] == 0).all()                      # the fixture's boards leave the two hidden rows empty, as every state does
    st = np.ascontiguousarray(x[:, 0, 2:, :].reshape(16, 200).astype(np.int8))
    out = np.zeros((16, 50), np.float32)
    oracle.lib().orc_distnet_forward(oracle.ptr(P), oracle.ptr(st), 16, 50, oracle.ptr(out))
    assert np.all(np.abs(out - g["y"]) <= 1e-6 * g["y"])
    assert np.allclose(out.sum(1), 1.0, atol=1e-6)
    assert np.allclose(np.log(out.astype(np.float64)), g["lp"], atol=2e-6)


# ---- agents/core_distributional.py (numba, fastmath): float tolerance, not bit patterns ----
RTOL, ATOL = 2e-6, 1e-7      # pure-Python run of the reference (float32 scalars under NumPy 2) vs numba's typing in double


def distpy_cases(golden_dir, prefix, keys):
    g = np.load(os.path.join(golden_dir, "ref_distpy.npz"))
    for i in range(int(g[prefix + "_n"])):
        yield {k: g["%s_%s_%d" % (prefix, k, i)] for k in keys}


def test_distpy_shift_matches_the_reference_function(oracle, golden_dir):
    L, n = oracle.lib(), 0
    for c in distpy_cases(golden_dir, "s", ("dist", "x", "vmax", "out")):
        d = np.ascontiguousarray(c["dist"], np.float32)
        out = np.zeros(len(d), np.float32)
        L.orc_distpy_shift(oracle.ptr(d), len(d), float(c["x"]), 0.0, float(c["vmax"]), oracle.ptr(out))
        assert np.allclose(out, c["out"], rtol=RTOL, atol=ATOL), (n, np.abs(out - c["out"]).max())
        assert abs(float(out.sum()) - float(d.sum())) < 1e-5      # a shift moves mass, it does not lose any
        n += 1
    assert n == 24


def test_distpy_policy_matches_the_reference_function(oracle, golden_dir):
    L, n = oracle.lib(), 0
    for c in distpy_cases(golden_dir, "p", ("stats", "nodes", "cur", "out")):
        ns = np.ascontiguousarray(c["stats"], np.float32)
        cn = np.ascontiguousarray(c["nodes"], np.int32)
        assert L.orc_distpy_policy(oracle.ptr(cn), len(cn), oracle.ptr(ns), float(c["cur"])) == int(c["out"]), n
        n += 1
    assert n == 40


def test_distpy_backup_matches_the_reference_function(oracle, golden_dir):
    L, n = oracle.lib(), 0
    for c in distpy_cases(golden_dir, "b", ("stats_in", "dist_in", "trace", "r", "leaf", "vmax", "stats_out", "dist_out")):
        ns = np.ascontiguousarray(c["stats_in"], np.float32).copy()
        nd = np.ascontiguousarray(c["dist_in"], np.float32).copy()
        tr = np.ascontiguousarray(c["trace"], np.int32)
        leaf = np.ascontiguousarray(c["leaf"], np.float32)
        bins = nd.shape[1]
        scratch = np.zeros(bins, np.float32)
        L.orc_distpy_backup(oracle.ptr(tr), len(tr), oracle.ptr(ns), oracle.ptr(nd), bins, float(c["r"]), oracle.ptr(leaf),
                            0.0, float(c["vmax"]), oracle.ptr(scratch))
        assert np.allclose(ns, c["stats_out"], rtol=1e-5, atol=1e-4), (n, np.abs(ns - c["stats_out"]).max())
        assert np.allclose(nd, c["dist_out"], rtol=RTOL, atol=ATOL), (n, np.abs(nd - c["dist_out"]).max())
        untouched = np.setdiff1d(np.arange(len(ns)), tr)
        assert ns[untouched].tobytes() == c["stats_in"][untouched].tobytes()
        n += 1
    assert n == 12


def test_oracle_dist_agent_invariants(oracle):
    """The oracle's DistValueSim (agent_oracle.c kind 6: TreeAgent's tree + the kernels above, which are pinned on the
    reference functions): visits add up, every visited node's distribution is normalised and its mean statistic is the
    running mean the backup defines; collections happen and keep the statistics of what stays reachable."""
    g = oracle.Game(seed=11)
    a = oracle.Agent(6, max_nodes=4000, low=5)
    a.update_root(g)
    sims = 120
    for m in range(14):
        act = a.play(sims)
        st = a.stats()
        assert 0 <= act < 7 and np.isfinite(st).all()
        assert act == int(np.argmax(st[1]))
        g.play(act)
        a.update_root(g)
    assert a.error == 0 and a.n_sims == 14 * sims and a.n_gc >= 1
    ns, nd = a.dist_arrays()
    arr = a.arrays()
    mark = np.zeros(4000, np.uint8)
    oracle.lib().orc_get_all_childs(a.root, oracle.ptr(arr["child"]), 4000, oracle.ptr(mark))
    occ = np.nonzero(mark)[0]
    occ = occ[occ != 0]
    vis = ns[occ, 0]
    assert (vis >= 0).all() and np.all(vis == np.round(vis))
    seen = occ[vis > 0]
    assert len(seen) > 50
    assert np.allclose(nd[seen].sum(1), 1.0, atol=2e-5)
    assert np.all(ns[occ, 2] == arr["score"][occ])            # node_stats[:, 2] is the node's score
    assert np.all(ns[seen, 3] >= 0)


def test_store_nodes_dist_follows_the_reference_text(oracle):
    """oracle kind 6's harvest (agent_oracle.c store_nodes_dist) against the reference's commented store_nodes
    (agents/DistValueSimOnline.py:116-141) applied literally, in Python, to the oracle's own arrays: `for idx in nodes: if
    ns[0] < min_visits or any(n_stats[child[idx][i]][0] < 1 ...): continue; states[m] = g_arr[idx].getState(); values[m] =
    n_dist[idx]; weights[m] = ns[0]` with nodes = the nodes a collection removes, ascending."""
    min_visits = 6
    g = oracle.Game(seed=2024)
    a = oracle.Agent(6, max_nodes=4000, low=5, online=True, min_visits_to_store=min_visits, memory_size=50000)
    a.update_root(g)
    checked = 0
    for m in range(40):
        act = a.play(120)
        g.play(act)
        a.update_root(g)
        if g.end:
            g.reset()
            a.update_root(g)
        # an explicit collection: what it stores must be what the reference's text selects from the state just before it
        ref = a.arrays()
        child, games = ref["child"].copy(), ref["games"].copy()
        ns, nd = (x.copy() for x in a.dist_arrays())
        mark = np.zeros(4000, np.uint8)
        oracle.lib().orc_get_all_childs(a.root, oracle.ptr(child), 4000, oracle.ptr(mark))
        nodes = [i for i in range(1, 4000) if not mark[i]]
        exp_states, exp_values, exp_weights = [], [], []
        for idx in nodes:
            if ns[idx][0] < min_visits or any(ns[child[idx][i]][0] < 1 for i in range(7)):
                continue
            st = np.zeros((20, 10), np.int8)
            oracle.lib().orc_game_render(oracle.ptr(games[idx:idx + 1]), oracle.ptr(st))
            exp_states.append(st.reshape(200)); exp_values.append(nd[idx]); exp_weights.append(ns[idx][0])
        before = a.memory_index
        a.remove_nodes()
        st, d, v = a.memory_dist()
        assert a.memory_index - before == len(exp_weights), (m, a.memory_index - before, len(exp_weights))
        if exp_weights:
            assert np.array_equal(st[before:], np.stack(exp_states)) and d[before:].tobytes() == np.stack(exp_values).tobytes()
            assert v[before:].tobytes() == np.asarray(exp_weights, np.float32).tobytes()
            checked += len(exp_weights)
    assert checked > 30
