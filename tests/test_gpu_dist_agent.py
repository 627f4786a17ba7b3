"""DistValueSim (TM_KIND_DIST, BASELINE configs[4]) on the GPU against the oracle's restatement of the same
agent (oracle/agent_oracle.c kind 6 = TreeAgent's tree + agents/core_distributional.py's kernels as oracle/dist_oracle.c
restates them, pinned on a pure-Python run of the reference functions).  Both sides evaluate leaves with the same hash
distribution, so actions, root statistics, per-node statistics and all 50-atom distributions must agree bit for bit -
through garbage collections too."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(z):
    z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M
    return z ^ (z >> np.uint64(31))


def hash_dist(states, bins=50):
    """oracle/agent_oracle.c orc_hash_dist in numpy: states int8 [k,20,10] -> float32 [k,bins]"""
    with np.errstate(over="ignore"):
        w = np.ascontiguousarray(states, np.int8).reshape(len(states), 200).view(np.uint64)      # [k,25]
        h = np.full(len(states), 0x9E3779B97F4A7C15, np.uint64)
        for j in range(25):
            h = _splitmix64(h ^ w[:, j])
        p = np.stack([((_splitmix64(h ^ np.uint64(b + 1)) & np.uint64(0xFFFF)) + np.uint64(1)).astype(np.float64) for b in range(bins)], 1)
    s = np.zeros(len(states))
    for b in range(bins):
        s = s + p[:, b]
    return (p / s[:, None]).astype(np.float32)


def test_hash_dist_matches_the_oracle(oracle):
    import ctypes as C
    rng = np.random.default_rng(1)
    st = rng.integers(-1, 2, size=(9, 200)).astype(np.int8)
    out = np.zeros((9, 50), np.float32)
    oracle.lib().orc_hash_dist(None, oracle.ptr(st), 9, 50, oracle.ptr(out))
    assert out.tobytes() == hash_dist(st.reshape(9, 20, 10)).tobytes()


@pytest.mark.parametrize("max_nodes,moves,sims", [(20000, 10, 120), (5000, 16, 150)])
def test_dist_agent_matches_the_oracle_bit_for_bit(oracle, max_nodes, moves, sims):
    import torch
    from tetris_mcts_amd import agents
    from tetris_mcts_amd.pyTetris import Tetris
    G = 6
    env_args = ((20, 10), 1, 0, 0)
    seeds = 4242 + np.arange(G)
    game = Tetris(*env_args, seed=seeds, n_games=G)
    agent = agents.DistValueSim(sims=sims, env=Tetris, env_args=env_args, n_games=G, max_nodes=max_nodes, evaluator=hash_dist)
    agent.update_root(game)
    og = [oracle.Game(seed=int(s)) for s in seeds]
    oa = [oracle.Agent(6, max_nodes=max_nodes, low=5) for _ in range(G)]
    for g in range(G):
        oa[g].update_root(og[g])
    for m in range(moves):
        act = np.atleast_1d(agent.play())
        stats = agent.get_stats().reshape(G, 3, 7)
        for g in range(G):
            a = oa[g].play(sims)
            assert a == act[g], ("action", m, g, a, act[g], oa[g].stats(), stats[g])
            assert oa[g].stats().tobytes() == stats[g].tobytes(), ("stats", m, g, oa[g].stats(), stats[g])
            og[g].play(a)
            oa[g].update_root(og[g])
        game.play(act)
        agent.update_root(game)
        ended = np.atleast_1d(game.end)
        if ended.any():
            game.reset("ended")
            agent.update_root(game)
            for g in np.nonzero(ended)[0]:
                og[g].reset()
                oa[g].update_root(og[g])
    s = agent.store
    assert (s.errors() == 0).all()
    gs = s.t["gs"].cpu().numpy()
    stat = s.t["obs_stat"].view(torch.float32).cpu().numpy()         # [G, N, 4] = visit, mean, variance, M2
    dist = s.t["node_dist"].cpu().numpy()
    n_gc = 0
    for g in range(G):
        assert oa[g].error == 0 and gs[g, 0] == oa[g].root and gs[g, 8] == oa[g].n_sims and gs[g, 7] == oa[g].n_expand
        assert gs[g, 9] == oa[g].n_gc
        n_gc += int(gs[g, 9])
        ns, nd = oa[g].dist_arrays()
        ref = oa[g].arrays()
        mark = np.zeros(max_nodes, np.uint8)
        oracle.lib().orc_get_all_childs(oa[g].root, oracle.ptr(ref["child"]), max_nodes, oracle.ptr(mark))
        occ = np.nonzero(mark)[0]
        occ = occ[occ != 0]
        assert stat[g, occ][:, [0, 1, 2, 3]].tobytes() == np.ascontiguousarray(ns[occ][:, [0, 1, 3, 4]]).tobytes(), g
        assert dist[g, occ, :50].tobytes() == np.ascontiguousarray(nd[occ]).tobytes(), g
        assert np.all(dist[g, occ, 50:] == 0)
    if max_nodes <= 5000:
        assert n_gc > 0          # the small pool went through collections


def _ref_net_params(golden_dir):
    import os
    g = np.load(os.path.join(golden_dir, "ref_distnet.npz"))
    keys = ["seq__conv1__weight", "seq__conv1__bias", "seq__conv2__weight", "seq__conv2__bias", "seq__fc1__weight",
            "seq__fc1__bias", "seq__fc_v__weight", "seq__fc_v__bias"]
    return g, np.concatenate([g[k].ravel() for k in keys]).astype(np.float32)


def test_hip_head_matches_the_oracle_bit_for_bit_and_the_reference_net(oracle, golden_dir):
    """csrc/distnet.hip (k_dn_conv + k_dn_fc, fp32 matrix cores) vs oracle/distnet_oracle.c: the same bits on random boards
    (a batch that is not a multiple of a tile, one state alone), and within 1e-6 relative of the reference's own Net on its
    fixture (model/model_distributional.py:18-57 run on CPU, tests/golden/ref_distnet.npz)."""
    import torch
    from tetris_mcts_amd.model_distributional import Model_Dist
    g, P = _ref_net_params(golden_dir)
    mdl = Model_Dist(atoms=50, backend="hip")
    mdl.set_flat_params(P)
    x = g["x"]
    assert (x[:, 0, :2] == 0).all()
    st = np.ascontiguousarray(x[:, 0, 2:, :].reshape(16, 200).astype(np.int8))
    y = mdl.inference_device(torch.from_numpy(st).cuda())[:, :50].cpu().numpy()
    assert np.abs(y - g["y"]).max() <= 1e-6 * np.abs(g["y"]).max() and np.all(np.abs(y - g["y"]) <= 1e-6 * g["y"] + 1e-9)
    assert np.array_equal(mdl.inference(x)[0], y)                      # the reference's signature ([B,1,22,10] floats)
    rng = np.random.default_rng(5)
    for n in (1, 16, 1003):
        st = rng.integers(-1, 2, size=(n, 200)).astype(np.int8)
        st[: n // 2, :100] = 0                                        # half of them with an empty upper half, like real boards
        ref = np.zeros((n, 50), np.float32)
        oracle.lib().orc_distnet_forward(oracle.ptr(P), oracle.ptr(st), n, 50, oracle.ptr(ref))
        out = mdl.inference_device(torch.from_numpy(st).cuda())
        assert out[:, :50].cpu().numpy().tobytes() == ref.tobytes(), n
        assert float(out[:, 50:].abs().sum()) == 0.0
    # other numbers of atoms go through the same kernels (padding rows of the last FC tile)
    for atoms in (7, 64):
        m2 = Model_Dist(atoms=atoms, backend="hip", seed=atoms)
        P2 = m2.flat_params().cpu().numpy()
        st = rng.integers(-1, 2, size=(40, 200)).astype(np.int8)
        ref = np.zeros((40, atoms), np.float32)
        oracle.lib().orc_distnet_forward(oracle.ptr(P2), oracle.ptr(st), 40, atoms, oracle.ptr(ref))
        out = m2.inference_device(torch.from_numpy(st).cuda())[:, :atoms].cpu().numpy()
        assert out.tobytes() == ref.tobytes(), atoms
        tor = Model_Dist(atoms=atoms, backend="torch", seed=atoms).inference_device(torch.from_numpy(st).cuda())[:, :atoms].cpu().numpy()
        assert np.allclose(out, tor, rtol=2e-5, atol=1e-8)


@pytest.mark.parametrize("max_nodes,moves,sims", [(20000, 6, 150), (4000, 12, 150)])
def test_dist_agent_with_the_hip_head_matches_the_oracle_bit_for_bit(oracle, max_nodes, moves, sims):
    """DistValueSim as it is benchmarked - the distributional head as HIP kernels inside the native launch loop (search.hip),
    requests rendered from the nodes' packed games - against oracle kind 6 with distnet_oracle.c as its evaluator: actions,
    root statistics, every reachable node's statistics and distributions, through collections."""
    import torch
    from tetris_mcts_amd import agents
    from tetris_mcts_amd.model_distributional import Model_Dist
    from tetris_mcts_amd.pyTetris import Tetris
    G = 8
    env_args = ((20, 10), 1, 0, 0)
    seeds = 777 + np.arange(G)
    model = Model_Dist(atoms=50, seed=0, backend="hip")
    P = model.flat_params().cpu().numpy()
    game = Tetris(*env_args, seed=seeds, n_games=G)
    agent = agents.DistValueSim(sims=sims, env=Tetris, env_args=env_args, n_games=G, max_nodes=max_nodes, model=model)
    assert agent.search_model() is model
    agent.update_root(game)
    og = [oracle.Game(seed=int(s)) for s in seeds]
    oa = [oracle.Agent(6, max_nodes=max_nodes, low=5, evaluator="distnet", params=P) for _ in range(G)]
    for g in range(G):
        oa[g].update_root(og[g])
    for m in range(moves):
        act = np.atleast_1d(agent.play())
        stats = agent.get_stats().reshape(G, 3, 7)
        for g in range(G):
            a = oa[g].play(sims)
            assert a == act[g], ("action", m, g, a, act[g], oa[g].stats(), stats[g])
            assert oa[g].stats().tobytes() == stats[g].tobytes(), ("stats", m, g, oa[g].stats(), stats[g])
            og[g].play(a)
            oa[g].update_root(og[g])
        game.play(act)
        agent.update_root(game)
        ended = np.atleast_1d(game.end)
        if ended.any():
            game.reset("ended")
            agent.update_root(game)
            for g in np.nonzero(ended)[0]:
                og[g].reset()
                oa[g].update_root(og[g])
    _compare_dist_trees(oracle, agent, oa, range(G), max_nodes)
    assert agent.store.search_stats(1, 0)["runs"] == moves             # the native loop ran the moves
    if max_nodes <= 4000:
        assert agent.store.counter("N_GC") > 0


def _compare_dist_trees(oracle, agent, oa, games, max_nodes, oracles=None):
    import torch
    s = agent.store
    assert (s.errors() == 0).all()
    gs = s.t["gs"].cpu().numpy()
    for i, g in enumerate(games):
        o = oa[i] if oracles is None else oracles[i]
        stat = s.t["obs_stat"][g].view(torch.float32).cpu().numpy()         # [N, 4] = visit, mean, variance, M2
        dist = s.t["node_dist"][g].cpu().numpy()
        assert o.error == 0 and gs[g, 0] == o.root and gs[g, 8] == o.n_sims and gs[g, 7] == o.n_expand, (g, gs[g, :10])
        assert gs[g, 9] == o.n_gc
        ns, nd = o.dist_arrays()
        ref = o.arrays()
        mark = np.zeros(max_nodes, np.uint8)
        oracle.lib().orc_get_all_childs(o.root, oracle.ptr(ref["child"]), max_nodes, oracle.ptr(mark))
        occ = np.nonzero(mark)[0]
        occ = occ[occ != 0]
        assert stat[occ][:, [0, 1, 2, 3]].tobytes() == np.ascontiguousarray(ns[occ][:, [0, 1, 3, 4]]).tobytes(), g
        assert dist[occ, :50].tobytes() == np.ascontiguousarray(nd[occ]).tobytes(), g
        assert np.all(dist[occ, 50:] == 0)


def test_sampled_games_of_the_benchmarked_dist_batch(oracle):
    """BASELINE configs[4] as bench.py runs it: 4096 games x 1000 sims/move, pool 100 000, Model_Dist(seed 0) on the HIP head
    through the native loop; games 0, 1337 and 4095 against their own oracles (kind 6 + distnet_oracle.c), three moves."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from tetris_mcts_amd import agents, dist as tdist
    from tetris_mcts_amd.model_distributional import Model_Dist
    from tetris_mcts_amd.pyTetris import Tetris
    G, sims, N, moves = 4096, 1000, 100000, 3
    env_args = ((20, 10), 1, 0, 0)
    seeds = np.asarray(tdist.game_seeds(20260925, G, 0))
    model = Model_Dist(atoms=50, seed=0, backend="hip")
    P = model.flat_params().cpu().numpy()
    game = Tetris(*env_args, seed=seeds, n_games=G)
    agent = agents.DistValueSim(sims=sims, env=Tetris, env_args=env_args, n_games=G, max_nodes=N, model=model)
    agent.update_root(game)
    sample = [0, 1337, 4095]
    og = [oracle.Game(seed=int(seeds[g])) for g in sample]
    oa = [oracle.Agent(6, max_nodes=N, low=5, evaluator="distnet", params=P) for _ in sample]
    for o, gm in zip(oa, og):
        o.update_root(gm)

    def omove(i):
        a = oa[i].play(sims)
        st = oa[i].stats()
        og[i].play(a)
        oa[i].update_root(og[i])
        assert not og[i].end
        return a, st

    with ThreadPoolExecutor(max_workers=len(sample)) as pool:
        for m in range(moves):
            fut = [pool.submit(omove, i) for i in range(len(sample))]
            act = np.atleast_1d(agent.play())
            stats = agent.get_stats().reshape(G, 3, 7)
            game.play(act)
            agent.update_root(game)
            assert not np.atleast_1d(game.end).any()
            for i, g in enumerate(sample):
                a, st = fut[i].result()
                assert a == act[g], ("action", m, g, a, act[g], st, stats[g])
                assert st.tobytes() == stats[g].tobytes(), ("stats", m, g, st, stats[g])
    gs = agent.store.t["gs"]
    assert (gs[:, 8] == moves * sims).all() and int(gs[:, 6].abs().sum().item()) == 0
    _compare_dist_trees(oracle, agent, oa, sample, N)
    del agent
    torch.cuda.empty_cache()


def test_dist_agent_with_the_distributional_net():
    """The reference's Net (model_distributional.py:18-57, mirrored on PyTorch-ROCm) as the evaluator: distributions stay
    normalised, statistics are finite, every game runs every simulation."""
    import torch
    from tetris_mcts_amd import agents
    from tetris_mcts_amd.model_distributional import Model_Dist
    from tetris_mcts_amd.pyTetris import Tetris
    G, sims = 64, 100
    env_args = ((20, 10), 1, 0, 0)
    game = Tetris(*env_args, seed=99, n_games=G)
    agent = agents.DistValueSim(sims=sims, env=Tetris, env_args=env_args, n_games=G, max_nodes=8000,
                                model=Model_Dist(atoms=50, seed=0, backend="torch"))
    assert agent.search_model() is False             # torch ops: the launch loop runs in TreeAgent.mcts
    agent.update_root(game)
    for m in range(3):
        act = agent.play()
        game.play(act)
        agent.update_root(game)
    s = agent.store
    gs = s.t["gs"]
    assert (gs[:, 8] == 3 * sims).all() and (s.errors() == 0).all()
    root = gs[:, 0].long()
    d = s.t["node_dist"][torch.arange(G, device=s.device), root, :50]
    assert torch.allclose(d.sum(1), torch.ones(G, device=s.device), atol=1e-4)
    mean, var = agent.get_value()
    assert np.isfinite(mean).all() and (np.asarray(var) >= -1e-6).all()


@pytest.mark.parametrize("max_nodes,sims,moves", [(3000, 150, 60), (1500, 60, 120)])
def test_dist_online_harvest_matches_the_oracle(oracle, max_nodes, sims, moves):
    """The online leg (DistValueSimOnline.store_nodes, agents/DistValueSimOnline.py:116-141): what the collections harvest on the
    device - freed nodes with enough visits whose seven children were all visited: board, 50-atom distribution, visit count, per
    game in index order, collection after collection - is what oracle kind 6 stores, bit for bit; the search itself too."""
    import torch
    from tetris_mcts_amd import agents, dist as tdist
    from tetris_mcts_amd.pyTetris import Tetris
    G = 6
    env_args = ((20, 10), 1, 0, 0)
    seeds = 31337 + np.arange(G)
    game = Tetris(*env_args, seed=seeds, n_games=G)
    agent = agents.DistValueSim(sims=sims, env=Tetris, env_args=env_args, n_games=G, max_nodes=max_nodes, evaluator=hash_dist,
                                online=True, min_visits_to_store=8, replay_cap=8192)
    agent.update_root(game)
    og = [oracle.Game(seed=int(s)) for s in seeds]
    oa = [oracle.Agent(6, max_nodes=max_nodes, low=5, online=True, min_visits_to_store=8, memory_size=100000) for _ in range(G)]
    for g in range(G):
        oa[g].update_root(og[g])
    for m in range(moves):
        act = np.atleast_1d(agent.play())
        stats = agent.get_stats().reshape(G, 3, 7)
        for g in range(G):
            a = oa[g].play(sims)
            assert oa[g].error == 0
            assert a == act[g] and oa[g].stats().tobytes() == stats[g].tobytes(), (m, g)
            og[g].play(a)
            oa[g].update_root(og[g])
        game.play(act)
        agent.update_root(game)
        ended = np.atleast_1d(game.end)
        if ended.any():
            game.reset("ended")
            agent.update_root(game)
            for g in np.nonzero(ended)[0]:
                og[g].reset()
                oa[g].update_root(og[g])
    s = agent.store
    assert (s.errors() == 0).all() and s.counter("N_DROPPED") == 0
    cnt = s.t["replay_count"].cpu().numpy()
    total = 0
    for g in range(G):
        st, d, v = oa[g].memory_dist()
        assert cnt[g] == len(v), (g, cnt[g], len(v))
        if cnt[g] == 0:
            continue
        keys = s.t["replay_obs"][g, :cnt[g]]
        dev_states = tdist.render_observations(keys).reshape(-1, 200).cpu().numpy().astype(np.int8)
        assert np.array_equal(dev_states, st), g
        assert s.t["replay_dist"][g, :cnt[g], :50].cpu().numpy().tobytes() == d.tobytes(), g
        assert s.t["replay_stat"][g, :cnt[g], 2].cpu().numpy().tobytes() == v.tobytes(), g
        total += int(cnt[g])
    assert total > 50 and s.counter("N_GC") == sum(o.n_gc for o in oa) > G
    # the agent's drain: the same tuples, the buffers emptied
    keys, dists, visits = agent.harvested()
    assert keys.shape[0] == total and int(s.t["replay_count"].sum().item()) == 0


def test_dist_online_training_round():
    """harvest -> train_nodes: the head is fitted on the harvested distributions (Model_Dist.loss through train.train_data), the
    HIP operand streams are rebuilt, and the search goes on with the new weights."""
    import torch
    from tetris_mcts_amd import agents
    from tetris_mcts_amd.model_distributional import Model_Dist
    from tetris_mcts_amd.pyTetris import Tetris
    G, sims = 32, 120
    env_args = ((20, 10), 1, 0, 0)
    game = Tetris(*env_args, seed=5, n_games=G)
    model = Model_Dist(atoms=50, seed=0, backend="hip")
    agent = agents.DistValueSim(sims=sims, env=Tetris, env_args=env_args, n_games=G, max_nodes=2500, model=model, online=True,
                                min_visits_to_store=8, memory_growth_rate=100)
    agent.update_root(game)
    res, before = None, model.flat_params().clone()
    for m in range(60):
        act = agent.play()
        game.play(act)
        agent.update_root(game)
        if game.end.any():
            game.reset("ended")
            agent.update_root(game)
        out = agent.train_if_collected(max_iters=60, iters_per_val=20, batch_size=256)
        if agent.n_trains >= 2:
            break
    assert agent.n_trains >= 2 and (agent.store.errors() == 0).all()
    assert not torch.equal(before, model.flat_params())
    d = model.inference_device(torch.zeros(4, 200, dtype=torch.int8, device="cuda"))[:, :50]
    assert torch.isfinite(d).all() and torch.allclose(d.sum(1), torch.ones(4, device="cuda"), atol=1e-5)
