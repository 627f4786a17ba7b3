"""Scheduling of the garbage collections by the collector workgroups of tm_sim_step (tree.hip gc_collector_block), beyond what
the parity tests of test_gpu_tree.py / test_gpu_benched_regime.py hold them to:
  * more games waiting than a launch looks after: the places go to the games that have waited longest, whatever their index;
  * a collection under way that is met by a launch with another number of collector workgroups is left alone and flagged;
  * the catch-up launches of the native loop run over the games that owe only (tm_store::game_list) - same results as the
    Python-driven loop, whose catch-up launches run over all games.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _agent(G, sims, max_nodes, seed, **kw):
    from gpu_helpers import hash_eval_torch
    from tetris_mcts_amd import agents
    from tetris_mcts_amd.pyTetris import Tetris
    game = Tetris((20, 10), 1, 0, 0, seed=seed, n_games=G)
    if "model" not in kw:
        kw["evaluator"] = hash_eval_torch
    agent = agents.ValueSim(sims=sims, env=Tetris, env_args=game.env_args, n_games=G, max_nodes=max_nodes, online=False, **kw)
    agent.update_root(game)
    return game, agent


def _step(game, agent):
    a = agent.play()
    game.play(a)
    agent.update_root(game)
    if np.atleast_1d(game.end).any():
        game.reset("ended")
        agent.update_root(game)
    return a


def test_waiting_games_are_served_by_age_not_by_index():
    """640 games whose 2000-node pools run dry within a few moves of each other: far more than 64 collections wait at a time.
    With the places given out in game-index order (rounds 3-4) the high-numbered games waited for all the others; by age the
    two ends of the batch wait about the same number of launches per collection."""
    from tetris_mcts_amd import store as st
    G = 640
    game, agent = _agent(G, 20, 2000, 4242)
    for _ in range(64):
        _step(game, agent)
    gs = agent.store.t["gs"].cpu().numpy()
    n_gc, waited = gs[:, st.GS["N_GC"]].astype(np.float64), gs[:, st.GS["GC_SLICES"]].astype(np.float64)
    assert n_gc.sum() >= 900 and (gs[:, st.GS["ERR"]] & ~1).max() == 0
    per = waited / np.maximum(n_gc, 1)
    first, last = per[:G // 10].mean(), per[-G // 10:].mean()
    print('launches waited per collection: first tenth %.2f, last tenth %.2f, all %.2f' % (first, last, per.mean()))
    assert 0.6 < last / first < 1.6, (first, last, per.mean())


def test_a_collection_met_by_another_grid_is_flagged_not_corrupted():
    """A collection is the work of launches with one number of collector workgroups (its shares and its arrival count are cut for
    that many).  Begun by launches over all 8 games (2 collector workgroups) and then met by launches over the two 4-game
    sub-batches (1 each), it is not touched, and the game carries TM_ERR_GC_GRID - the agent raises instead of going on with
    corrupt free lists (round-3 advisor)."""
    from tetris_mcts_amd import store as st
    from tetris_mcts_amd.model import Model_VV
    game, agent = _agent(8, 40, 1500, 99, model=Model_VV(backend="hip", seed=0), n_sub=1)
    S = agent.store
    both = st.SIM_BACKUP | st.SIM_FRONT
    found = False
    for move in range(60):
        # the Python-driven protocol over the WHOLE store, launch by launch, until a collection is under way
        S.move_begin(40)
        S.sim_step(both)
        for _ in range(40):
            agent.evaluate_requests()
            S.sim_step(both)
            ph = (S.t["gs"][:, st.GS["GC_PHASE"]] & 15).cpu().numpy()
            if ((ph >= 2) & (ph <= 6)).any():
                found = True
                break
        if found:
            break
        todo, collecting = S.sims_remaining()
        while todo:
            for _ in range(todo):
                agent.evaluate_requests()
                S.sim_step(both)
            todo, collecting = S.sims_remaining()
            while collecting:
                S.gc_step()
                todo, collecting = S.sims_remaining()
        _, action = S.root_stats()
        game.play(action.cpu().numpy())
        agent.update_root(game)
        if np.atleast_1d(game.end).any():
            game.reset("ended")
            agent.update_root(game)
    assert found, "no collection was caught under way"
    ph0 = S.t["gs"][:, st.GS["GC_PHASE"]] & 15
    mid = (ph0 >= 2) & (ph0 <= 6)
    under_way = (mid | (ph0 == 9) | (ph0 == 10)).cpu().numpy()      # ... or a speculative marking the whole-store launches began
    free_before = S.t["free_node"][mid].clone()
    # ... and now the native loop with two sub-batches: launches over 4 games each
    with pytest.raises(RuntimeError, match="tree engine error flags"):
        for _ in range(3):
            S.search(5, agent.model, n_sub=2)
            agent.get_action()
    err = S.errors().cpu().numpy()
    assert (err[mid.cpu().numpy()] & 16).all() and not (err[~under_way] & 16).any()
    assert torch.equal(S.t["free_node"][mid], free_before)          # the collection was left where it stood


def test_catch_up_over_the_owing_games_equals_catch_up_over_all_games():
    """The native loop's catch-up launches run over the games tm_sims_owing lists; the Python-driven loop's over the whole grid.
    96 games with 3000-node pools (a few collections under way on most moves), the same evaluator: every action and every root
    statistic of every game is the same, and the native loop's catch-up launches were small."""
    from tetris_mcts_amd.model import Model_VV
    G, sims, moves = 96, 60, 45
    model = Model_VV(backend="hip", seed=0)
    game_a, native = _agent(G, sims, 3000, 321, model=model)
    game_b, python = _agent(G, sims, 3000, 321, model=model)
    python.search_model = lambda: False          # TreeAgent.mcts drives the launches itself (evaluate_requests -> the same HIP net)
    for m in range(moves):
        a, b = _step(game_a, native), _step(game_b, python)
        assert np.array_equal(a, b), m
        assert native.get_stats().tobytes() == python.get_stats().tobytes(), m
    ss = native.store.search_stats(native.n_sub, native.ev_every, reset=False)
    assert native.store.counter("N_GC") == python.store.counter("N_GC") >= G
    assert ss["catchup_launches"] > 0 and ss["catchup_waves"] < 0.5 * G * ss["catchup_launches"]


def test_marker_with_the_bitmaps_in_memory(oracle, monkeypatch):
    """Pools of more than 100 000 nodes do not fit the marking workgroups' LDS: the marker then keeps its three bitmaps in memory
    (tree.hip gc_sweep_mark<false>: the same sweeps on device-scope atomics).  TM_GC_MARKS_IN_MEMORY=1 makes every launch take
    that form, so that it can be held to the oracle on pools the oracle replays in seconds: six games with one marking workgroup
    between them, and forty games with five (several owners per game, cross-share sends), every move of every game, trees
    included, through their collections."""
    from test_gpu_tree import _compare_run
    monkeypatch.setenv("TM_GC_MARKS_IN_MEMORY", "1")
    assert _compare_run(oracle, "ValueSim", G=6, sims=40, max_nodes=8000, seed=11, moves=110, evaluator="hash", check_tree_every=25) >= 1
    assert _compare_run(oracle, "ValueSimLP", G=40, sims=30, max_nodes=6000, seed=5, moves=90, evaluator="hash", check_tree_every=30) >= 10


def test_every_phase_word_has_its_summary_bit():
    """The collectors find the collecting games through TM_GS_GC_ACTIVE4 (a bit a game in the control block of every fourth game)
    instead of reading every game's phase word: a phase without its bit would never be served.  Launch by launch over a batch
    whose small pools collect all the time (requests, speculative markings, collections under way, resumes, root moves that drop
    markings, pools outgrown and reset): after every launch every non-zero phase word has its bit, and the native loop (two
    sub-batches: slices of the store) keeps it so."""
    from tetris_mcts_amd import store as st
    from tetris_mcts_amd.model import Model_VV
    G = 72
    game, agent = _agent(G, 40, 1500, 7, model=Model_VV(backend="hip", seed=0), n_sub=1)
    S = agent.store
    both = st.SIM_BACKUP | st.SIM_FRONT
    idx = torch.arange(G, device=S.device)

    def check(where):
        gs = S.t["gs"]
        phase = gs[:, st.GS["GC_PHASE"]]
        bit = (gs[(idx & ~3), st.GS["GC_ACTIVE4"]] >> (idx & 3)) & 1
        bad = (phase != 0) & (bit == 0)
        assert not bool(bad.any()), (where, torch.nonzero(bad).flatten().tolist(), phase[bad].tolist())
        return int((phase != 0).sum())

    seen = 0
    for move in range(40):
        S.move_begin(40)
        S.sim_step(both)
        for _ in range(40):
            agent.evaluate_requests()
            S.sim_step(both)
            seen += check("launch of move %d" % move)
        todo, collecting = S.sims_remaining()
        while todo or collecting:
            while collecting:
                S.gc_step()
                check("collector-only launch")
                todo, collecting = S.sims_remaining()
            for _ in range(todo):
                agent.evaluate_requests()
                S.sim_step(both)
                check("catch-up launch")
            todo, collecting = S.sims_remaining()
        _, action = S.root_stats()
        game.play(action.cpu().numpy())
        agent.update_root(game)
        check("update_root")
        if np.atleast_1d(game.end).any():
            game.reset("ended")
            agent.update_root(game)
    assert seen > 500 and int(S.t["gs"][:, st.GS["N_GC"]].sum()) > 100, (seen, int(S.t["gs"][:, st.GS["N_GC"]].sum()))
    assert (S.errors().cpu().numpy() & ~1).max() == 0
    # ... and through the native loop with two sub-batches (slices of 36 games)
    for _ in range(12):
        S.search(40, agent.model, n_sub=2)
        check("native loop")
        a = agent.get_action()
        game.play(a)
        agent.update_root(game)
        if np.atleast_1d(game.end).any():
            game.reset("ended")
            agent.update_root(game)
