"""Parity in the regime the headline number is measured in (BASELINE configs[1]/[2]): 500 simulations per move, the full
100 000-entry pool, the benchmark's own network weights (Net() under manual_seed(0)) and game seeds (20260925 + g) -
HIP engine vs per-game CPU oracles, bit for bit (action, 3x7 root statistics, whole-tree export), through the native
launch loop (search.hip) with sub-batches on separate streams.

Covered on purpose, with assertions that they really happened:
  * traces longer than 128 nodes (two flushes of the 64-entry LDS trace buffer, three chunks of the backup),
  * a garbage collection at the full 100 000-entry pool, by the collector workgroups, with the catch-up launches that follow,
  * sampled games of a real 4096-game batch (results must not depend on the batch a game runs in),
  * the steady state the second bench window is taken in: the same 4096-game batch played on to move 90, a thousand
    collections by the collector workgroups under load (speculative marking, waiting lists, catch-up launches), games
    chosen AFTER the run by what happened to them - incl. one whose reachable tree outgrew the pool.
The oracle games run in parallel threads (liboracle releases the GIL)."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
BASE_SEED = 20260925
KIND = {"ValueSim": 0, "ValueSimLP": 1, "ValueSimC": 2}


def _bench_model():
    from tetris_mcts_amd.model import Model_VV
    model = Model_VV(backend="hip", seed=0)      # exactly bench.py's network
    return model, model.flat_params().cpu().numpy()


class _OracleGame:
    def __init__(self, oracle, kind, seed, params, max_nodes):
        self.g = oracle.Game(seed=seed)
        self.a = oracle.Agent(kind, max_nodes=max_nodes, evaluator="valuenet", params=params)
        self.a.update_root(self.g)

    def move(self, sims):
        act = self.a.play(sims)
        stats = self.a.stats()
        self.g.play(act)
        self.a.update_root(self.g)
        ended = self.g.end
        if ended:
            self.g.reset()
            self.a.update_root(self.g)
        return act, stats, ended


def _run(oracle, name, seeds, sample, moves, sims=500, max_nodes=100000, n_sub=1, check_tree=True, checkpoint=None, **agent_kw):
    """Batch of len(seeds) games on the GPU; games `sample` (indices) are checked against their own oracle every move."""
    from tetris_mcts_amd import agents
    from tetris_mcts_amd.pyTetris import Tetris
    model, params = _bench_model()
    if checkpoint:
        model.load(checkpoint, verbose=False)
        params = model.flat_params().cpu().numpy()
    env_args = ((20, 10), 1, 0, 0)
    seeds = np.asarray(seeds, dtype=np.int64)
    G = len(seeds)
    game = Tetris(*env_args, seed=seeds, n_games=G)
    agent = getattr(agents, name)(sims=sims, env=Tetris, env_args=env_args, n_games=G, max_nodes=max_nodes, model=model,
                                  online=False, n_sub=n_sub, **agent_kw)
    agent.update_root(game)
    orc = [_OracleGame(oracle, KIND[name], int(seeds[g]), params, max_nodes) for g in sample]
    with ThreadPoolExecutor(max_workers=len(sample)) as pool:
        for m in range(moves):
            fut = [pool.submit(o.move, sims) for o in orc]       # the oracles work while the GPU does
            act = np.atleast_1d(agent.play())
            stats = agent.get_stats().reshape(G, 3, 7)
            game.play(act)
            agent.update_root(game)
            ended = np.atleast_1d(game.end)
            if ended.any():
                game.reset("ended")
                agent.update_root(game)
            for g, f in zip(sample, fut):
                a, s, e = f.result()
                assert a == act[g], (name, "move", m, "game", g, a, act[g], s, stats[g])
                assert s.tobytes() == stats[g].tobytes(), (name, "stats", m, g, s, stats[g])
                assert bool(e) == bool(ended[g]), (name, "end", m, g)
    st = agent.store
    assert (st.errors() == 0).all()
    for g, o in zip(sample, orc):
        assert o.a.error == 0
        gs = st.t["gs"][g].cpu().numpy()
        assert gs[9] == o.a.n_gc and gs[7] == o.a.n_expand and gs[8] == o.a.n_sims, (g, gs[:10], o.a.n_gc, o.a.n_expand)
        assert gs[19] == o.a.max_trace_len, (g, gs[19], o.a.max_trace_len)
        if check_tree:
            dev = st.export_game(g)
            ref = o.a.arrays()
            mark = np.zeros(max_nodes, np.uint8)
            oracle.lib().orc_get_all_childs(o.a.root, oracle.ptr(ref["child"]), max_nodes, oracle.ptr(mark))
            occ = np.nonzero(mark)[0]
            assert gs[0] == o.a.root
            for k in ("child", "score", "n_to_o"):
                assert np.array_equal(dev[k][occ], ref[k][occ]), (k, g)
            oo = np.unique(ref["n_to_o"][occ])
            for k in ("visit", "value", "variance", "end_obs"):
                assert dev[k][oo].tobytes() == ref[k][oo].tobytes(), (k, g)
    return agent, orc


@pytest.mark.parametrize("name,moves", [("ValueSim", 20), ("ValueSimLP", 12), ("ValueSimC", 12)])
def test_500_sims_full_pool_sampled_seeds(oracle, name, moves):
    """Four of the benchmark's games (g = 0, 1, 1337, 4095) as a batch of 4, two sub-batches on two streams."""
    seeds = [BASE_SEED + g for g in (0, 1, 1337, 4095)]
    agent, orc = _run(oracle, name, seeds, sample=[0, 1, 2, 3], moves=moves, n_sub=1 if name == "ValueSimC" else 2)
    assert int(agent.store.t["gs"][:, 19].max().item()) > 64       # the LDS trace buffer was flushed at least once


def test_sampled_games_under_the_trained_checkpoint(oracle):
    """The regime self-play lives in: the value net bench.py's `trained_net` object runs under (the committed checkpoint of an
    online run; outputs scaled to [0, 13 508] x [0.1, 2.5e7] instead of the random-init net's [0, 100] x [0.1, 1000]) - every
    simulation expands a node, pools fill within ten moves and the trees that survive a move are large.  256 games x 500
    simulations, 30 000-node pools, 26 moves: three games against their oracles through their collections, whole trees."""
    import os
    import bench
    ck = os.path.join(bench.ROOT, bench.TRAINED_CHECKPOINT)
    seeds = BASE_SEED + np.arange(256)
    agent, orc = _run(oracle, "ValueSim", seeds, sample=[0, 100, 255], moves=26, max_nodes=30000, checkpoint=ck)
    gs = agent.store.t["gs"].cpu().numpy()
    assert gs[:, 9].sum() > 256 and min(o.a.n_gc for o in orc) >= 1          # collections everywhere, in the checked games too
    assert gs[:, 7].sum() > 0.95 * gs[:, 8].sum()                            # (nearly) every simulation expands


def test_traces_beyond_128_and_a_collection_at_the_full_pool(oracle):
    """Game 1337 of the benchmark walks 130-node traces around move 50; game 1 exhausts its 100 000-entry pool at move 74
    (86 000 nodes reachable at that collection).  Both, 76 moves, every move compared."""
    agent, orc = _run(oracle, "ValueSim", [BASE_SEED + 1337, BASE_SEED + 1], sample=[0, 1], moves=76, n_sub=1)
    gs = agent.store.t["gs"].cpu().numpy()
    assert gs[0, 19] > 128, gs[0, 19]                 # TM_GS_MAX_TRACE: > 2 LDS flushes, 3 backup chunks
    assert gs[1, 9] >= 1                              # TM_GS_N_GC: the collection happened ...
    assert gs[1, 24] > 50000                          # ... with most of the pool reachable
    assert gs[1, 38] >= 4 * gs[1, 9]                  # TM_GS_GC_SLICES: ... over several launches (the steps of tree.hip)
    assert gs[1, 37] >= gs[1, 9]                      # TM_GS_GC_MARK_LAUNCHES: ... marked by the collector's sweep marker
    ss = agent.store.search_stats(1, 0)
    assert ss["catchup_launches"] > 0                 # the collecting game caught up after the regular launches


@pytest.mark.parametrize("name,moves,n_sub", [("ValueSim", 6, 4), ("ValueSimLP", 4, 1)])
def test_sampled_games_of_a_real_4096_game_batch(oracle, name, moves, n_sub):
    """BASELINE configs[1] and configs[2] themselves: 4096 games x 500 sims/move (ValueSim: four sub-batches; ValueSimLP:
    28 672 request slots, seven per game, one batch); games 0, 1337 and 4095 against their own oracles (a game's results
    must not depend on the batch or the sub-batch it runs in)."""
    import torch
    seeds = BASE_SEED + np.arange(4096)
    agent, orc = _run(oracle, name, seeds, sample=[0, 1337, 4095], moves=moves, n_sub=n_sub, check_tree=True)
    gs = agent.store.t["gs"]
    assert (gs[:, 8] == moves * 500).all()           # every game ran every simulation
    assert int(gs[:, 6].abs().sum().item()) == 0
    if name == "ValueSimLP":
        assert agent.store.eval_slots == 7 and agent.store.t["eval_obs"].numel() == 28672
        # several unique children per expansion; only those on their first visit are posted to the net (TM_SIM_EVAL_NEEDED)
        posted, skipped, expanded = int(gs[:, 17].sum().item()), int(gs[:, 48].sum().item()), int(gs[:, 7].sum().item())
        assert posted + skipped > 3 * expanded and skipped > 0 and posted > expanded
        used = sum(o.a.n_eval_used for o in orc)
        assert sum(int(gs[g, 17].item()) for g in [0, 1337, 4095]) == used          # exactly the ones the backup uses
        assert sum(int(gs[g, 17].item() + gs[g, 48].item()) for g in [0, 1337, 4095]) == sum(o.a.n_eval_states for o in orc)
    else:
        # ValueSim: a leaf whose observation was evaluated before (under the same weights) is answered from obs_eval
        assert sum(int(gs[g, 49].item()) for g in [0, 1337, 4095]) == sum(o.a.n_eval_repeat for o in orc)
        assert sum(int(gs[g, 17].item() + gs[g, 49].item()) for g in [0, 1337, 4095]) == sum(o.a.n_eval_states for o in orc)
    del agent
    torch.cuda.empty_cache()


def test_steady_state_of_the_real_4096_game_batch(oracle):
    """The regime of bench.py's steady-state window (moves 76-95): the benchmark's 4096 games x 500 sims x 100 000-node pools
    played to move 90 through the native loop.  Every move's actions and root statistics of ALL games are recorded; the games
    to check are picked afterwards - game 1 (collects at move 74), the games with the most collections, one that waited
    longest for its collection, and one whose reachable tree outgrew the pool (the reference's MAX_NODES EXCEEDED state,
    agent.py:96-97 / agent.cpp:227-231: compared up to the move the oracle reports the same exhaustion in, and the restart
    must happen exactly there) - and replayed by their own oracles in threads: actions, statistics bytes, episode ends,
    collection / expansion / simulation counters and the whole reachable tree."""
    import torch
    from tetris_mcts_amd import agents
    from tetris_mcts_amd.pyTetris import Tetris
    MOVES, SIMS, N, G = 90, 500, 100000, 4096
    model, params = _bench_model()
    env_args = ((20, 10), 1, 0, 0)
    seeds = BASE_SEED + np.arange(G)
    game = Tetris(*env_args, seed=seeds, n_games=G)
    agent = agents.ValueSim(sims=SIMS, env=Tetris, env_args=env_args, n_games=G, max_nodes=N, model=model, online=False)
    agent.update_root(game)
    st = agent.store
    acts = np.zeros((MOVES, G), np.int32)
    stats = np.zeros((MOVES, G, 21), np.float32)
    ends = np.zeros((MOVES, G), bool)
    resets = np.zeros((MOVES, G), np.int32)          # TM_GS_N_POOL_RESET after the move's update_root
    for m in range(MOVES):
        acts[m] = np.atleast_1d(agent.play())
        stats[m] = agent.get_stats().reshape(G, 21)
        game.play(acts[m])
        agent.update_root(game)
        ends[m] = np.atleast_1d(game.end)
        if ends[m].any():
            game.reset("ended")
            agent.update_root(game)
        resets[m] = st.t["gs"][:, 18].cpu().numpy()
    gs = st.t["gs"].cpu().numpy()
    n_gc, n_reset = gs[:, 9], gs[:, 18]
    assert int((gs[:, 6] & ~1).sum()) == 0                     # no error flag but the pool one
    assert (gs[:, 8] == MOVES * SIMS).all()                    # every game ran every simulation (catch-up launches included)
    total_gc = int(n_gc.sum())
    ss = st.search_stats(1, 0)
    print("collections", total_gc, "trees restarted", int(n_reset.sum()), "catch-up launches", ss["catchup_launches"],
          "collector-only launches", ss["gc_launches"])
    assert total_gc >= 1000, total_gc                          # the concurrent regime, not a collection here and there
    assert ss["catchup_launches"] > 0
    clean = n_reset == 0
    order = np.argsort(-(n_gc * clean), kind="stable")
    sample = [1] + [int(g) for g in order[:2] if g != 1]
    slices_per_gc = np.where(n_gc > 0, gs[:, 38] / np.maximum(n_gc, 1), 0) * clean
    slow = int(np.argmax(slices_per_gc))
    if slow not in sample:
        sample.append(slow)
    full = np.nonzero(n_reset > 0)[0]
    if len(full):
        first_reset = (resets[:, full] > 0).argmax(0)
        sample.append(int(full[np.argmin(first_reset)]))       # the earliest restart: most moves after it are not comparable anyway
    assert clean[1] and n_gc[1] >= 1

    def replay(g):
        o = _OracleGame(oracle, 0, int(seeds[g]), params, N)
        for m in range(MOVES):
            a, s, e = o.move(SIMS)
            if o.a.error:
                return o, m                                    # the oracle's pool is exhausted in move m (MAX_NODES EXCEEDED)
            assert a == acts[m, g], ("action", g, m, a, acts[m, g])
            assert s.tobytes() == stats[m, g].tobytes(), ("stats", g, m, s, stats[m, g])
            assert bool(e) == bool(ends[m, g]), ("end", g, m)
        return o, MOVES

    with ThreadPoolExecutor(max_workers=len(sample)) as pool:
        res = list(pool.map(replay, sample))
    for g, (o, upto) in zip(sample, res):
        if n_reset[g] > 0:
            # the tree was restarted at the update_root of the very move in which the reference runs out of nodes
            assert upto < MOVES and o.a.error == 1, (g, upto)
            assert resets[upto, g] >= 1 and (upto == 0 or resets[upto - 1, g] == 0), (g, upto, resets[:, g])
            continue
        assert upto == MOVES and o.a.error == 0
        assert gs[g, 9] == o.a.n_gc and gs[g, 7] == o.a.n_expand and gs[g, 8] == o.a.n_sims, (g, gs[g, :10], o.a.n_gc)
        assert gs[g, 19] == o.a.max_trace_len
        dev = st.export_game(g)
        ref = o.a.arrays()
        mark = np.zeros(N, np.uint8)
        oracle.lib().orc_get_all_childs(o.a.root, oracle.ptr(ref["child"]), N, oracle.ptr(mark))
        occ = np.nonzero(mark)[0]
        assert gs[g, 0] == o.a.root
        for k in ("child", "score", "n_to_o"):
            assert np.array_equal(dev[k][occ], ref[k][occ]), (k, g)
        oo = np.unique(ref["n_to_o"][occ])
        for k in ("visit", "value", "variance", "end_obs"):
            assert dev[k][oo].tobytes() == ref[k][oo].tobytes(), (k, g)
    assert sum(int(n_gc[g]) for g in sample if n_reset[g] == 0) >= 3
    del agent
    torch.cuda.empty_cache()
