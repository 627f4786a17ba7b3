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
    agent = agents.DistValueSim(sims=sims, env=Tetris, env_args=env_args, n_games=G, max_nodes=8000, model=Model_Dist(atoms=50, seed=0))
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
