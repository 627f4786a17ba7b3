"""Pin oracle/agent_oracle.c against the reference's own Python agents.

tests/golden/ref_agents.json was produced by agents/ValueSim.py and agents/ValueSimLP.py imported
unmodified from /root/reference (tests/golden/make_golden.py): per move (action, score, lines, 3x7 stats
bytes), across GC events (small pools) and episode resets.
"""
import json
import os

import numpy as np
import pytest

KIND = {"ValueSim": 0, "ValueSimLP": 1}


def _runs(golden_dir):
    with open(os.path.join(golden_dir, "ref_agents.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("idx", [0, 1, 2, 3])
def test_agent_oracle_replays_reference(oracle, golden_dir, idx):
    r = _runs(golden_dir)[idx]
    params = np.load(os.path.join(golden_dir, "ref_valuenet.npz"))["params"]
    g = oracle.Game(seed=r["seed"])
    a = oracle.Agent(KIND[r["name"]], max_nodes=r["max_nodes"], evaluator=r["evaluator"], params=params)
    a.update_root(g)
    for i, (act, score, lines, stats_hex) in enumerate(r["moves"]):
        got = a.play(r["sims"])
        assert a.error == 0
        assert got == act, (i, got, act)
        assert a.stats().astype("<f4").tobytes().hex() == stats_hex, i
        g.play(got)
        a.update_root(g)
        assert (g.score, g.line_clears) == (score, lines), i
        if g.end:
            g.reset()
            a.update_root(g)
    if r["max_nodes"] < 100000:
        assert a.n_gc >= 1  # the fixture crosses at least one pool-exhaustion GC
    assert a.sweep_mismatch == 0     # the device collector's marking schedule, run beside get_all_childs at every collection
    a.close()


def test_sweep_marker_schedule_finds_the_reference_set_at_every_collection(oracle):
    """oracle/uct_oracle.c orc_sweep_marks (the schedule of tree.hip's collector: descending sweeps over a pending bitmap)
    against orc_get_all_childs (core.h:32-50, pinned on the reference's core.cpp) at every collection of long games with small
    pools - trees that have been through many collections, where index order no longer follows allocation order."""
    for kind, seed, pool in ((0, 5, 3000), (1, 6, 6000), (0, 7, 30000)):
        g = oracle.Game(seed=seed)
        a = oracle.Agent(kind, max_nodes=pool, evaluator="hash")
        a.update_root(g)
        for _ in range(400 if pool < 30000 else 150):
            g.play(a.play(60 if pool < 30000 else 400))
            a.update_root(g)
            if g.end:
                g.reset()
                a.update_root(g)
        assert a.n_gc >= (10 if pool < 30000 else 1), a.n_gc
        assert a.sweep_mismatch == 0
        # every reachable node's row is read once; a handful of sweeps, whatever the depth of the tree
        assert a.sweep_max_passes <= 12, a.sweep_max_passes
        a.close()


def test_store_nodes_emits_replay_tuples(oracle):
    g = oracle.Game(seed=21)
    a = oracle.Agent(0, max_nodes=6000, online=True, memory_size=100000, min_visits_to_store=3)
    a.update_root(g)
    for _ in range(120):
        g.play(a.play(40))
        a.update_root(g)
        if g.end:
            g.reset()
            a.update_root(g)
    assert a.n_gc >= 1
    st, val, var, vis = a.memory()
    assert len(st) > 0 and (vis >= 3).all()
    assert set(np.unique(st)) <= {-1, 0, 1}
    a.close()


@pytest.mark.parametrize("idx", [0, 1])
def test_agent_oracle_replays_reference_cpp_agent(oracle, golden_dir, idx):
    """tests/golden/ref_cppagent.json: agents/cppmodule/agent.cpp MCTSAgent (compiled in place), LP and non-LP."""
    with open(os.path.join(golden_dir, "ref_cppagent.json")) as f:
        r = json.load(f)[idx]
    g = oracle.Game(seed=r["seed"])
    a = oracle.Agent(2 if r["lp"] else 3, max_nodes=r["max_nodes"], cpp_occupied=True)
    a.update_root(g)
    for i, (act, score, lines) in enumerate(r["moves"]):
        got = a.play(r["sims"])
        assert got == act, (i, got, act)
        g.play(got)
        a.update_root(g)
        assert (g.score, g.line_clears) == (score, lines), i
        if g.end:
            g.reset()
            a.update_root(g)
    assert a.n_gc >= 1 and a.sweep_mismatch == 0
    a.close()


@pytest.mark.parametrize("idx", [0, 1])
def test_agent_oracle_replays_reference_vanilla(oracle, golden_dir, idx):
    """tests/golden/ref_vanilla.json: agents/Vanilla.py (random rollouts via random.randint) - BASELINE configs[0]."""
    import random
    with open(os.path.join(golden_dir, "ref_vanilla.json")) as f:
        r = json.load(f)[idx]
    g = oracle.Game(seed=r["seed"])
    a = oracle.Agent(4, max_nodes=r["max_nodes"], gamma=0.99, low=5)
    a.set_python_random_state(random.Random(r["random_seed"]).getstate())
    a.update_root(g)
    for i, (act, score, lines, stats_hex) in enumerate(r["moves"]):
        got = a.play(r["sims"])
        assert got == act, (i, got, act)
        assert a.stats().astype("<f4").tobytes().hex() == stats_hex, i
        g.play(got)
        a.update_root(g)
        assert (g.score, g.line_clears) == (score, lines), i
        if g.end:
            g.reset()
            a.update_root(g)
    a.close()


def test_mt19937_randint_matches_cpython(oracle):
    import random
    for seed in (0, 5, 123456789):
        r = random.Random(seed)
        st = np.asarray(r.getstate()[1], dtype=np.uint64).astype(np.uint32)
        got = [oracle.lib().orc_mt_randint7_test(oracle.ptr(st)) for _ in range(2000)]
        assert got == [r.randint(0, 6) for _ in range(2000)]


@pytest.mark.parametrize("idx", [0, 1])
def test_store_nodes_matches_reference_python_agents(oracle, golden_dir, idx):
    """tests/golden/ref_online_py.json: what the reference's ValueSim / ValueSimLP (online=True) put into agent.memory at
    every GC (store_nodes, ValueSim.py:122-159) - states, values, variances, visit weights, in order."""
    import hashlib
    with open(os.path.join(golden_dir, "ref_online_py.json")) as f:
        r = json.load(f)[idx]
    g = oracle.Game(seed=r["seed"])
    a = oracle.Agent({"ValueSim": 0, "ValueSimLP": 1}[r["name"]], max_nodes=r["max_nodes"], online=True,
                     memory_size=1 << 18, min_visits_to_store=r["min_visits_to_store"])
    a.update_root(g)
    got, seen, n_gc = [], 0, 0
    for m, act in enumerate(r["actions"]):
        assert a.play(r["sims"]) == act, m
        if a.n_gc != n_gc:
            n_gc = a.n_gc
            st, val, var, vis = a.memory()
            got.append(dict(move=m, size=len(val) - seen, states_sha1=hashlib.sha1(st[seen:].tobytes()).hexdigest(),
                            value=val[seen:].astype("<f4").tobytes().hex(), variance=var[seen:].astype("<f4").tobytes().hex(),
                            visit=vis[seen:].astype("<f4").tobytes().hex()))
            seen = len(val)
        g.play(act)
        a.update_root(g)
        if g.end:
            g.reset()
            a.update_root(g)
    assert len(got) >= 3 and got == r["harvests"]
    a.close()


@pytest.mark.parametrize("idx", [0, 1, 2])
def test_agent_oracle_replays_reference_other_environments(oracle, golden_dir, idx):
    """tests/golden/ref_agents_env.json: ValueSim (app 2, scoring 1, uniform randomizer), ValueSimLP (app 3) and the
    compiled C++ MCTSAgent (app 2, uniform randomizer) of the reference on the oracle engine: with gravity every `app`
    actions a trace can hold the same observation twice, the case where the backup must not be reordered."""
    with open(os.path.join(golden_dir, "ref_agents_env.json")) as f:
        r = json.load(f)[idx]
    app, scoring, randomizer = r["env"]
    cpp = r["name"] == "MCTSAgent"
    g = oracle.Game(app, scoring, randomizer, r["seed"])
    a = oracle.Agent(2 if cpp else KIND[r["name"]], max_nodes=r["max_nodes"], app=app, scoring=scoring,
                     randomizer=randomizer, cpp_occupied=cpp)
    a.update_root(g)
    for i, (act, score, lines, stats_hex) in enumerate(r["moves"]):
        got = a.play(r["sims"])
        assert a.error == 0
        assert got == act, (i, got, act)
        if stats_hex:
            assert a.stats().astype("<f4").tobytes().hex() == stats_hex, i
        g.play(got)
        a.update_root(g)
        assert (g.score, g.line_clears) == (score, lines), i
        if g.end:
            g.reset()
            a.update_root(g)
    a.close()


@pytest.mark.parametrize("idx", [0, 1, 2])
def test_agent_oracle_replays_reference_vanillac(oracle, golden_dir, idx):
    """tests/golden/ref_vanillac.json: the reference's VanillaC = the compiled MCTSAgent with evaluator type 1
    (agent.cpp:447-455) and agents/VanillaC.py's random playout (randint(0, 7), variance 1e5), gamma 0.99."""
    import random
    with open(os.path.join(golden_dir, "ref_vanillac.json")) as f:
        r = json.load(f)[idx]
    g = oracle.Game(seed=r["seed"])
    a = oracle.Agent(5, max_nodes=r["max_nodes"], gamma=0.99, low=1, cpp_occupied=True)   # the compiled agent's GC slip
    a.set_python_random_state(random.Random(r["random_seed"]).getstate())
    a.update_root(g)
    for i, (act, score, lines) in enumerate(r["moves"]):
        got = a.play(r["sims"])
        assert got == act, (i, got, act)
        g.play(got)
        a.update_root(g)
        assert (g.score, g.line_clears) == (score, lines), i
        if g.end:
            g.reset()
            a.update_root(g)
    assert a.n_gc >= (1 if idx == 2 else 0)
    a.close()


def treeagent_script(steps, script_seed):
    """tests/golden/make_golden.py:treeagent_script (the golden stores the seed, not the actions)."""
    rs = np.random.RandomState(script_seed)
    return [[int(rs.randint(7)), int(rs.randint(7))] for _ in range(steps)]


def replay_treeagent(agent, game, make_game, script, on_step):
    """Drive an agent through the steps of tests/golden/ref_treeagent.json with its single calls.  `agent` offers
    update_root / expand / new_node, games offer play / copy_from / end / reset; on_step(i, kids, gk, root, ended)."""
    tmp, tmp2 = make_game(), make_game()
    agent.update_root(game)
    for i, (a1, a) in enumerate(script):
        agent.expand(game)
        kids = []
        for b in range(7):
            tmp.copy_from(game)
            tmp.play(b)
            kids.append(agent.new_node(tmp))
        tmp.copy_from(game)
        tmp.play(a1)
        agent.expand(tmp)
        gk = []
        for b in range(7):
            tmp2.copy_from(tmp)
            tmp2.play(b)
            gk.append(agent.new_node(tmp2))
        game.play(a)
        agent.update_root(game)
        root = agent.new_node(game)
        ended = game.end
        if np.any(ended):
            game.reset(ended) if np.ndim(ended) else game.reset()
            agent.update_root(game)
        on_step(i, kids, gk, root, ended)


@pytest.mark.parametrize("idx", [0, 1])
def test_agent_oracle_replays_reference_treeagent_single_calls(oracle, golden_dir, idx):
    """tests/golden/ref_treeagent.json: the reference's compiled TreeAgent answering update_root / expand(game) /
    new_node(game) (agent.cpp:825-833) - every index it returned, across collections at a 1000-node pool (run 1)."""
    with open(os.path.join(golden_dir, "ref_treeagent.json")) as f:
        r = json.load(f)[idx]
    a = oracle.Agent(3, max_nodes=r["max_nodes"], cpp_occupied=True)
    g = oracle.Game(seed=r["seed"])

    def check(i, kids, gk, root, ended):
        want = r["steps"][i]
        assert [kids, gk, root, int(ended)] == want, (i, kids, gk, root, ended, want)
    replay_treeagent(a, g, lambda: oracle.Game(seed=0), treeagent_script(len(r["steps"]), r["script_seed"]), check)
    assert a.error == 0
    assert (a.n_gc >= 2) if idx == 1 else (a.n_gc == 0)
