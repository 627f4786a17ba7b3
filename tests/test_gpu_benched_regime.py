"""Parity in the regime the headline number is measured in (BASELINE configs[1]/[2]): 500 simulations per move, the full
100 000-entry pool, the benchmark's own network weights (Net() under manual_seed(0)) and game seeds (20260925 + g) -
HIP engine vs per-game CPU oracles, bit for bit (action, 3x7 root statistics, whole-tree export), through the native
launch loop (search.hip) with sub-batches on separate streams.

Covered on purpose, with assertions that they really happened:
  * traces longer than 128 nodes (two flushes of the 64-entry LDS trace buffer, three chunks of the backup),
  * a garbage collection at the full 100 000-entry pool, by the collector workgroups, with the catch-up launches that follow,
  * sampled games of a real 4096-game batch (results must not depend on the batch a game runs in).
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


def _run(oracle, name, seeds, sample, moves, sims=500, max_nodes=100000, n_sub=1, check_tree=True, **agent_kw):
    """Batch of len(seeds) games on the GPU; games `sample` (indices) are checked against their own oracle every move."""
    from tetris_mcts_amd import agents
    from tetris_mcts_amd.pyTetris import Tetris
    model, params = _bench_model()
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


def test_traces_beyond_128_and_a_collection_at_the_full_pool(oracle):
    """Game 1337 of the benchmark walks 130-node traces around move 50; game 1 exhausts its 100 000-entry pool at move 74
    (86 000 nodes reachable at that collection).  Both, 76 moves, every move compared."""
    agent, orc = _run(oracle, "ValueSim", [BASE_SEED + 1337, BASE_SEED + 1], sample=[0, 1], moves=76, n_sub=1)
    gs = agent.store.t["gs"].cpu().numpy()
    assert gs[0, 19] > 128, gs[0, 19]                 # TM_GS_MAX_TRACE: > 2 LDS flushes, 3 backup chunks
    assert gs[1, 9] >= 1                              # TM_GS_N_GC: the collection happened ...
    assert gs[1, 24] > 50000                          # ... with most of the pool reachable
    assert gs[1, 38] > 5 * gs[1, 9]                   # TM_GS_GC_SLICES: ... over several launches (the steps of tree.hip)
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
        assert int(gs[:, 17].sum().item()) > 3 * int(gs[:, 7].sum().item())     # several evaluated states per expansion
    del agent
    torch.cuda.empty_cache()
