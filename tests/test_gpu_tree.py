"""The HIP tree engine (tree.hip) vs the oracle agents, bit for bit: actions, scores, root statistics, whole
tree arrays, across pool-exhaustion GCs and episode resets; and vs the reference's own golden runs."""
import json
import os

import numpy as np
import pytest

from gpu_helpers import hash_eval_torch

pytestmark = pytest.mark.gpu
KIND = {"ValueSim": 0, "ValueSimLP": 1, "ValueSimC": 2}


def _make(name, G, sims, max_nodes, seed, evaluator=None, model=None, **kw):
    from tetris_mcts_amd import agents
    from tetris_mcts_amd.pyTetris import Tetris
    env_args = kw.pop("env_args", ((20, 10), 1, 0, 0))
    game = Tetris(*env_args, seed=seed, n_games=G)
    agent = getattr(agents, name)(sims=sims, env=Tetris, env_args=env_args, n_games=G, max_nodes=max_nodes,
                                  evaluator=evaluator, model=model, online=kw.pop("online", False), **kw)
    agent.update_root(game)
    return game, agent


def _compare_run(oracle, name, G, sims, max_nodes, seed, moves, evaluator, params=None, model=None, env_args=None,
                 check_tree_every=0, okind=None, out=None, **agent_kw):
    env_args = env_args or ((20, 10), 1, 0, 0)
    game, agent = _make(name, G, sims, max_nodes, seed, evaluator=hash_eval_torch if evaluator == "hash" else None,
                        model=model, env_args=env_args, **agent_kw)
    og = [oracle.Game(env_args[1], env_args[2], env_args[3], seed + g) for g in range(G)]
    oa = [oracle.Agent(KIND[name] if okind is None else okind, max_nodes=max_nodes, app=env_args[1], scoring=env_args[2],
                       randomizer=env_args[3], evaluator=evaluator, params=params) for _ in range(G)]
    for g in range(G):
        oa[g].update_root(og[g])
    for m in range(moves):
        act = np.atleast_1d(agent.play())
        stats = agent.get_stats().reshape(G, 3, 7)
        for g in range(G):
            a = oa[g].play(sims)
            assert oa[g].error == 0
            assert a == act[g], (name, "move", m, "game", g, a, act[g], oa[g].stats(), stats[g])
            assert oa[g].stats().tobytes() == stats[g].tobytes(), (name, "stats", m, g, oa[g].stats(), stats[g])
            og[g].play(a)
            oa[g].update_root(og[g])
        game.play(act)
        agent.update_root(game)
        assert [o.score for o in og] == list(np.atleast_1d(game.score))
        if check_tree_every and (m + 1) % check_tree_every == 0:
            for g in (0, G - 1):
                dev = agent.store.export_game(g)
                ref = oa[g].arrays()
                mark = np.zeros(max_nodes, np.uint8)
                oracle.lib().orc_get_all_childs(oa[g].root, oracle.ptr(ref["child"]), max_nodes, oracle.ptr(mark))
                occ = np.nonzero(mark)[0]
                assert agent.store.t["gs"][g, 0].item() == oa[g].root
                for k in ("child", "score", "n_to_o"):
                    assert np.array_equal(dev[k][occ], ref[k][occ]), (k, m, g)
                oo = np.unique(ref["n_to_o"][occ])
                for k in ("visit", "value", "variance", "end_obs"):
                    assert dev[k][oo].tobytes() == ref[k][oo].tobytes(), (k, m, g)
        ended = np.atleast_1d(game.end)
        if ended.any():
            game.reset("ended")
            for g in np.nonzero(ended)[0]:
                og[g].reset()
            agent.update_root(game)
            for g in np.nonzero(ended)[0]:
                oa[g].update_root(og[g])
    gcs = sum(o.n_gc for o in oa)
    assert agent.store.counter("N_GC") == gcs
    assert agent.store.counter("N_EXPAND") == sum(o.n_expand for o in oa)
    if out is not None:
        out.update(agent=agent, oracles=oa)
    return gcs


def test_valuesim_hash_eval_with_gc(oracle):
    gcs = _compare_run(oracle, "ValueSim", G=6, sims=40, max_nodes=8000, seed=11, moves=150, evaluator="hash",
                       check_tree_every=25)
    assert gcs >= 1


def test_valuesimlp_hash_eval_with_gc(oracle):
    gcs = _compare_run(oracle, "ValueSimLP", G=6, sims=30, max_nodes=8000, seed=12, moves=150, evaluator="hash",
                       check_tree_every=25)
    assert gcs >= 1


@pytest.mark.parametrize("gc_spec_nodes", [None, 0, 2000])
def test_many_games_collecting_at_once(oracle, gc_spec_nodes):
    """The collector workgroups under load: 96 games with 3000-node pools, every game collects every few dozen moves and many
    of them at the same time (shared marking workgroups, bounded steps queueing for their cost allowance, oldest request
    first), every action and every root statistic against the games' own oracles, whole trees every 30 moves.  With the
    default speculative marking (a game with fewer than 256 free nodes is marked while it simulates), without it (0), and
    with every game that can run dry in its move being marked (2000: dozens of speculative markings at a time, requests
    overtaken by the pool running dry)."""
    gcs = _compare_run(oracle, "ValueSim", G=96, sims=30, max_nodes=3000, seed=77, moves=150, evaluator="hash",
                       check_tree_every=30, gc_spec_nodes=gc_spec_nodes)
    assert gcs >= 300


@pytest.mark.parametrize("gc_spec_nodes", [None, 2000])
def test_more_collections_at_once_than_the_collectors_list(oracle, gc_spec_nodes):
    """640 games with equal 2000-node pools run dry within a few moves of each other: hundreds of collections under way at
    the same time - more than the collector workgroups look after per launch (64 waiting games, 192 speculative markings:
    the others wait their turn), more marking games than marking workgroups (their time is shared)."""
    gcs = _compare_run(oracle, "ValueSim", G=640, sims=20, max_nodes=2000, seed=4242, moves=64, evaluator="hash",
                       check_tree_every=32, gc_spec_nodes=gc_spec_nodes)
    assert gcs >= 900


@pytest.mark.parametrize("name", ["ValueSimLP", "ValueSimC"])
def test_leaf_parallel_agents_collecting_under_load(oracle, name):
    """The leaf-parallel kinds (ValueSimLP: core.h:303-381 numerics; ValueSimC: agent.cpp:517-566, float carry, end_obs) through
    the collector workgroups under load: 256 games with 6000-node pools, dozens of collections under way at a time, seven
    request slots per game; every action and every root statistic of every game, whole trees every 25 moves."""
    gcs = _compare_run(oracle, name, G=256, sims=30, max_nodes=6000, seed=9090, moves=140, evaluator="hash", check_tree_every=35)
    assert gcs >= 256


def test_valuesim_app2_uniform_randomizer(oracle):
    _compare_run(oracle, "ValueSim", G=5, sims=30, max_nodes=20000, seed=5, moves=60, evaluator="hash",
                 env_args=((20, 10), 2, 1, 1))


def test_valuesimlp_app3(oracle):
    _compare_run(oracle, "ValueSimLP", G=5, sims=25, max_nodes=20000, seed=6, moves=50, evaluator="hash",
                 env_args=((20, 10), 3, 0, 0))


@pytest.mark.parametrize("name,sims,moves", [("ValueSim", 25, 12), ("ValueSimLP", 10, 8)])
def test_value_net_in_the_loop(oracle, golden_dir, name, sims, moves):
    from tetris_mcts_amd.model import Model_VV
    params = np.load(os.path.join(golden_dir, "ref_valuenet.npz"))["params"]
    model = Model_VV(backend="hip")
    model.set_flat_params(params)
    _compare_run(oracle, name, G=4, sims=sims, max_nodes=100000, seed=13, moves=moves, evaluator="valuenet",
                 params=params, model=model)


@pytest.mark.parametrize("max_nodes,moves", [(100000, 10), (6000, 40)])
def test_cpp_single_leaf_agent_in_the_native_loop(oracle, golden_dir, max_nodes, moves):
    """The all-C++ agent without leaf parallelism (agent.cpp:437-446,496-513; TM_KIND_CPPAGENT) with the HIP value net through the
    native launch loop - the other kind whose leaves are answered from the per-observation cache (TM_SIM_EVAL_NEEDED) - against
    oracle kind 3, which evaluates every leaf: same actions, statistics and trees, and as many cache answers as the oracle
    counts repeated observations; with a small pool through collections (a freed observation's cache entry is reset when the
    slot is handed out again)."""
    from tetris_mcts_amd.model import Model_VV
    params = np.load(os.path.join(golden_dir, "ref_valuenet.npz"))["params"]
    model = Model_VV(backend="hip")
    model.set_flat_params(params)
    out = {}
    gcs = _compare_run(oracle, "ValueSimC", G=4, sims=60, max_nodes=max_nodes, seed=77, moves=moves, evaluator="valuenet",
                       params=params, model=model, okind=3, out=out, leaf_parallel=False, check_tree_every=moves)
    agent, oa = out["agent"], out["oracles"]
    assert agent.search_model() is model and agent.store.kind == 3
    cached, posted = agent.store.counter("N_EVAL_CACHED"), agent.store.counter("N_EVAL")
    assert cached == sum(o.n_eval_repeat for o in oa) and cached > 0
    assert posted + cached == sum(o.n_eval_states for o in oa)
    if max_nodes < 10000:
        assert gcs >= 4


@pytest.mark.parametrize("idx", [0, 1, 2, 3])
def test_reference_golden_runs(golden_dir, idx):
    """tests/golden/ref_agents.json: the reference's own Python agents (make_golden.py)."""
    from tetris_mcts_amd.model import Model_VV
    with open(os.path.join(golden_dir, "ref_agents.json")) as f:
        r = json.load(f)[idx]
    model = None
    if r["evaluator"] == "valuenet":
        model = Model_VV(backend="hip")
        model.set_flat_params(np.load(os.path.join(golden_dir, "ref_valuenet.npz"))["params"])
    game, agent = _make(r["name"], 1, r["sims"], r["max_nodes"], r["seed"],
                        evaluator=hash_eval_torch if r["evaluator"] == "hash" else None, model=model)
    for i, (act, score, lines, stats_hex) in enumerate(r["moves"]):
        got = agent.play()
        assert got == act, (i, got, act)
        assert agent.get_stats().astype("<f4").tobytes().hex() == stats_hex, i
        game.play(got)
        agent.update_root(game)
        assert (game.score, game.line_clears) == (score, lines), i
        if game.end:
            game.reset()
            agent.update_root(game)


def test_full_size_batch_properties():
    """4096 games x 500 sims (BASELINE config 2 sizes, hash evaluator): size-independent invariants."""
    import torch
    G, sims = 4096, 500
    game, agent = _make("ValueSim", G, sims, 12000, 20260925, evaluator=hash_eval_torch)
    agent.play()
    st = agent.store
    gs = st.t["gs"].cpu().numpy()
    assert (gs[:, 6] == 0).all()                      # no error flags
    assert (gs[:, 8] == sims).all()                   # every game ran every simulation
    stats = agent.get_stats()
    # each simulation passes through the root exactly once: root observation visits == sims
    root = gs[:, 0]
    rec = st.t["node_rec"][torch.arange(G, device="cuda"), torch.as_tensor(root, device="cuda").long()]
    root_obs = rec[:, 29].long()
    visits = st.t["obs_stat"][torch.arange(G, device="cuda"), root_obs, 0].cpu().numpy()
    assert (visits == sims).all()
    # the unique children of the root absorb all but the first simulation
    assert (stats[:, 0].sum(axis=1) >= sims - 1).all()
    # nodes allocated == pool size - free, and never more than 7 per expansion + the root
    used = (12000 - 1) - gs[:, 2]
    assert (used <= 7 * gs[:, 7] + 1).all() and (used >= gs[:, 7]).all()


@pytest.mark.parametrize("idx", [0, 1])
def test_reference_cpp_agent_golden_runs(golden_dir, idx):
    """tests/golden/ref_cppagent.json: the reference's all-C++ MCTSAgent (agent.cpp compiled in place)."""
    with open(os.path.join(golden_dir, "ref_cppagent.json")) as f:
        r = json.load(f)[idx]
    game, agent = _make("ValueSimC", 1, r["sims"], r["max_nodes"], r["seed"], evaluator=hash_eval_torch,
                        leaf_parallel=r["lp"])
    for i, (act, score, lines) in enumerate(r["moves"]):
        got = agent.play()
        assert got == act, (i, got, act)
        game.play(got)
        agent.update_root(game)
        assert (game.score, game.line_clears) == (score, lines), i
        if game.end:
            game.reset()
            agent.update_root(game)
    assert agent.store.counter("N_GC") >= 1


def test_replay_harvest_matches_oracle(oracle):
    """GC-time replay tuples (ValueSim.store_nodes, ValueSim.py:122-159): same multiset as the oracle's memory."""
    G, sims, mn = 3, 40, 6000
    game, agent = _make("ValueSim", G, sims, mn, 21, evaluator=hash_eval_torch, online=True, min_visits_to_store=3,
                        replay_cap=20000)
    og = [oracle.Game(seed=21 + g) for g in range(G)]
    oa = [oracle.Agent(0, max_nodes=mn, online=True, memory_size=100000, min_visits_to_store=3) for _ in range(G)]
    for g in range(G):
        oa[g].update_root(og[g])
    for m in range(120):
        act = agent.play()
        for g in range(G):
            a = oa[g].play(sims)
            assert a == act[g]
            og[g].play(a)
            oa[g].update_root(og[g])
        game.play(act)
        agent.update_root(game)
        ended = game.end
        if ended.any():
            game.reset("ended")
            agent.update_root(game)
            for g in np.nonzero(ended)[0]:
                og[g].reset()
                oa[g].update_root(og[g])
    assert agent.store.counter("N_GC") == sum(o.n_gc for o in oa) >= 1
    cnt = agent.store.t["replay_count"].cpu().numpy()
    for g in range(G):
        st_o, val_o, var_o, vis_o = oa[g].memory()
        assert cnt[g] == len(val_o)
        keys = agent.store.t["replay_obs"][g, :cnt[g]].cpu().numpy().view(np.uint32)
        stat = agent.store.t["replay_stat"][g, :cnt[g]].cpu().numpy()
        # render the packed observations with the oracle and compare tuple by tuple (same ascending order)
        for i in range(cnt[g]):
            o = np.zeros(1, oracle.OBS_DTYPE)
            o.view(np.uint32)[:] = keys[i]
            out = np.zeros(200, np.int8)
            oracle.lib().orc_obs_render(oracle.ptr(o), oracle.ptr(out))
            assert np.array_equal(out, st_o[i]), (g, i)
        assert stat[:, 0].tobytes() == val_o.tobytes() and stat[:, 1].tobytes() == var_o.tobytes()
        assert np.array_equal(stat[:, 2], vis_o)


def test_online_training_loop(tmp_path, monkeypatch):
    """ValueSim online: GC harvests (state, TD target) tuples on the device, train_nodes fits the net on them
    (Yogi + Gaussian KL, ValueSim.py:161-185, model.py:176-249) and the search keeps running with the new weights."""
    import torch
    from tetris_mcts_amd import model as M
    monkeypatch.setattr(M, "EXP_PATH", str(tmp_path) + "/")
    model = M.Model_VV(backend="hip", seed=0)
    game, agent = _make("ValueSim", 16, 40, 5000, 5, model=model, online=True, min_visits_to_store=3, replay_cap=8192)
    before = model.flat_params().clone()
    for m in range(90):
        act = agent.play()
        game.play(act)
        agent.update_root(game)
        if game.end.any():
            game.reset("ended")
            agent.update_root(game)
    assert agent.store.counter("N_GC") >= 1
    n_tuples = int(agent.store.t["replay_count"].sum().item())
    assert n_tuples > 50
    res = agent.train_nodes(iters_per_val=20, batch_size=256, max_iters=60, dump_data=True,
                            dump_path=str(tmp_path / "data" / "dump"))
    assert res is not None and res["iters"] >= 20
    dump = np.load(str(tmp_path / "data" / "dump.npz"))       # the reference's replay dump layout (ValueSim.py:176-177)
    assert sorted(dump.files) == ["states", "values", "variance", "weights"]
    assert dump["states"].shape == (n_tuples, 1, 20, 10) and dump["weights"].shape == (n_tuples, 1)
    assert set(np.unique(dump["states"])) <= {-1.0, 0.0, 1.0} and (dump["weights"] >= 3).all()
    after = model.flat_params()
    assert not torch.equal(before, after)
    assert int(agent.store.t["replay_count"].sum().item()) == 0
    act = agent.play()            # the search runs on with the updated network (prepared weight streams refreshed)
    assert act.shape == (16,) and (agent.store.errors() == 0).all()


@pytest.mark.parametrize("idx", [0, 1])
def test_reference_vanilla_golden_runs(golden_dir, idx):
    """tests/golden/ref_vanilla.json: the reference's own Vanilla agent (BASELINE configs[0]: 100 sims/move, rollouts)."""
    from tetris_mcts_amd import agents
    from tetris_mcts_amd.pyTetris import Tetris
    with open(os.path.join(golden_dir, "ref_vanilla.json")) as f:
        r = json.load(f)[idx]
    game = Tetris((20, 10), 1, 0, 0, seed=r["seed"], n_games=1)
    agent = agents.Vanilla(sims=r["sims"], env=Tetris, env_args=game.env_args, n_games=1, max_nodes=r["max_nodes"],
                           random_seed=r["random_seed"])
    agent.update_root(game)
    for i, (act, score, lines, stats_hex) in enumerate(r["moves"]):
        got = agent.play()
        assert got == act, (i, got, act)
        assert agent.get_stats().astype("<f4").tobytes().hex() == stats_hex, i
        game.play(got)
        agent.update_root(game)
        assert (game.score, game.line_clears) == (score, lines), i
        if game.end:
            game.reset()
            agent.update_root(game)


def test_vanilla_batch_vs_oracle(oracle):
    import random
    from tetris_mcts_amd import agents
    from tetris_mcts_amd.pyTetris import Tetris
    G, sims, mn = 6, 30, 6000
    game = Tetris((20, 10), 1, 0, 0, seed=50, n_games=G)
    agent = agents.Vanilla(sims=sims, env=Tetris, env_args=game.env_args, n_games=G, max_nodes=mn, random_seed=9)
    agent.update_root(game)
    og = [oracle.Game(seed=50 + g) for g in range(G)]
    oa = [oracle.Agent(4, max_nodes=mn, gamma=0.99, low=5) for _ in range(G)]
    for g in range(G):
        oa[g].set_python_random_state(random.Random(9 + g).getstate())
        oa[g].update_root(og[g])
    for m in range(80):
        act = agent.play()
        stats = agent.get_stats()
        for g in range(G):
            a = oa[g].play(sims)
            assert a == act[g], (m, g)
            assert oa[g].stats().tobytes() == stats[g].tobytes(), (m, g)
            og[g].play(a)
            oa[g].update_root(og[g])
        game.play(act)
        agent.update_root(game)
        ended = game.end
        if ended.any():
            game.reset("ended")
            agent.update_root(game)
            for g in np.nonzero(ended)[0]:
                og[g].reset()
                oa[g].update_root(og[g])
    assert agent.store.counter("N_GC") == sum(o.n_gc for o in oa)


def test_pool_exhaustion_resets_the_tree_instead_of_undefined_behaviour():
    """A pool smaller than the reachable tree: the reference prints MAX_NODES EXCEEDED and runs into UB
    (agent.cpp:227-231); here the affected games restart with an empty tree and play on."""
    game, agent = _make("ValueSim", 8, 80, 400, 3, evaluator=hash_eval_torch)
    for m in range(40):
        act = agent.play()
        game.play(act)
        agent.update_root(game)
        if game.end.any():
            game.reset("ended")
            agent.update_root(game)
    assert agent.store.counter("N_POOL_RESET") > 0
    assert (agent.store.errors() == 0).all()
    strict = _make("ValueSim", 2, 80, 400, 3, evaluator=hash_eval_torch, reset_on_pool_exhaustion=False)[1]
    with pytest.raises(RuntimeError):
        for m in range(40):
            strict.play()
            strict.store.sim_step(0)


@pytest.mark.parametrize("idx", [0, 1, 2, 3])
def test_online_cpp_agent_training_sets(oracle, golden_dir, idx):
    """ValueSimC(online=True) under the four accumulation policies of OnlineMCTSAgent (agent.cpp:619-816): every
    training set handed to train() - device GC harvest -> ReplayMemory (drop / trim / policy) -> rendered states - is
    identical to the oracle agent + oracle memory on the same run, and the 900 actions are the ones the reference's
    compiled OnlineMCTSAgent played (tests/golden/ref_online_cpp.json).  The oracle runs WITHOUT the reference's
    `occupied`-vector slip here: the product implements the intended GC, so its training sets
    differ from the golden ones by the few tuples that slip withholds - pinned separately in tests/test_oracle_replay.py."""
    import hashlib
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import replay_oracle as ro
    with open(os.path.join(golden_dir, "ref_online_cpp.json")) as f:
        r = json.load(f)[idx]
    got_calls, want_calls, move = [], [], [0]

    def payload(st, val, var, vis):
        return dict(move=move[0], size=len(val), states_sha1=hashlib.sha1(st.astype(np.int8).tobytes()).hexdigest(),
                    value=val.astype("<f4").tobytes().hex(), variance=var.astype("<f4").tobytes().hex(),
                    visit=vis.astype("<f4").tobytes().hex())

    def train(state, value, variance, visit, size):
        assert state.shape == (size, 1, 20, 10)
        got_calls.append(payload(state.cpu().numpy().reshape(size, 200), value.cpu().numpy().ravel(),
                                 variance.cpu().numpy().ravel(), visit.cpu().numpy().ravel()))
    game, agent = _make("ValueSimC", 1, r["sims"], r["max_nodes"], r["seed"], evaluator=hash_eval_torch, online=True,
                        accumulation_policy=r["policy"], memory_size=r["memory_size"],
                        episodes_per_train=r["episodes_per_train"], memory_growth_rate=r["growth"],
                        min_visit=r["min_visit"], train=train, replay_cap=r["max_nodes"])
    og = oracle.Game(seed=r["seed"])
    oa = oracle.Agent(2, max_nodes=r["max_nodes"], online=True, min_visits_to_store=r["min_visit"], memory_size=1 << 20)
    oa.update_root(og)
    mem = ro.OnlineMemory(r["policy"], r["memory_size"], r["episodes_per_train"], r["growth"])
    seen, n_gc = 0, 0
    # (policies 0 and 3 - one episode-driven, one memory-driven - replay the reference's whole 900-move run; 1 and 2 its first
    # 500 moves: three to four trainings each, at half the suite time.  TM_TEST_FULL=1 - the round's evidence run, scripts/gpu_r06.sh
    # suite - replays all 900 moves under every policy: >= 7 collections and >= 4 trainings each)
    full = os.environ.get("TM_TEST_FULL", "0") == "1"
    n_moves = len(r["actions"]) if (full or r["policy"] in (0, 3)) else 500
    for m, act in enumerate(r["actions"][:n_moves]):
        move[0] = m
        got = agent.play()
        assert got == act == oa.play(r["sims"]), m
        if oa.n_gc != n_gc:
            n_gc = oa.n_gc
            st, val, var, vis = oa.memory()
            batch = [(st[i], val[i], var[i], vis[i]) for i in range(seen, len(val))]
            seen = len(val)
            out = mem.remove_nodes(batch, oa.episode)
            if out is not None:
                want_calls.append(payload(np.stack([e[0] for e in out]), np.asarray([e[1] for e in out]),
                                          np.asarray([e[2] for e in out]), np.asarray([e[3] for e in out])))
        assert agent.memory.memory_index == mem.memory_index, m
        game.play(got)
        agent.update_root(game)
        og.play(got)
        oa.update_root(og)
        if game.end:
            game.reset()
            agent.update_root(game)
            og.reset()
            oa.update_root(og)
    full = n_moves == len(r["actions"])
    assert agent.store.counter("N_GC") == oa.n_gc >= (7 if full else 3)
    assert len(want_calls) >= (4 if full else 2) and got_calls == want_calls
    if r["policy"] in (0, 1):
        # episode-driven policies train at the same moves as the reference's own run (the slip only changes how many
        # tuples a GC yields, which moves the memory-driven policies 2 and 3)
        assert [c["move"] for c in got_calls] == [c["move"] for c in r["train_calls"] if c["move"] < n_moves]


@pytest.mark.parametrize("idx", [0, 1])
def test_reference_vanillac_golden_runs(golden_dir, idx):
    """tests/golden/ref_vanillac.json: the reference's VanillaC = compiled MCTSAgent, evaluator type 1 (agent.cpp:447-455),
    agents/VanillaC.py's playout (randint(0, 7), variance 1e5, gamma 0.99).  Runs 0 and 1 stay inside their pools; run 2 of
    the file goes through collections, where the compiled agent's `occupied` slip changes what
    survives - that one is replayed by the oracle with the slip modelled (tests/test_oracle_agent.py), and the product is
    checked through collections against the oracle without it (test_vanillac_batch_vs_oracle)."""
    from tetris_mcts_amd import agents
    from tetris_mcts_amd.pyTetris import Tetris
    with open(os.path.join(golden_dir, "ref_vanillac.json")) as f:
        r = json.load(f)[idx]
    game = Tetris((20, 10), 1, 0, 0, seed=r["seed"], n_games=1)
    agent = agents.VanillaC(sims=r["sims"], env=Tetris, env_args=game.env_args, n_games=1, max_nodes=r["max_nodes"],
                            random_seed=r["random_seed"])
    agent.update_root(game)
    for i, (act, score, lines) in enumerate(r["moves"]):
        got = agent.play()
        assert got == act, (i, got, act)
        game.play(got)
        agent.update_root(game)
        assert (game.score, game.line_clears) == (score, lines), i
        if game.end:
            game.reset()
            agent.update_root(game)


def test_vanillac_batch_vs_oracle(oracle):
    import random
    from tetris_mcts_amd import agents
    from tetris_mcts_amd.pyTetris import Tetris
    G, sims, mn = 6, 30, 5000
    game = Tetris((20, 10), 1, 0, 0, seed=60, n_games=G)
    agent = agents.VanillaC(sims=sims, env=Tetris, env_args=game.env_args, n_games=G, max_nodes=mn, random_seed=3)
    agent.update_root(game)
    og = [oracle.Game(seed=60 + g) for g in range(G)]
    oa = [oracle.Agent(5, max_nodes=mn, gamma=0.99, low=1) for _ in range(G)]
    for g in range(G):
        oa[g].set_python_random_state(random.Random(3 + g).getstate())
        oa[g].update_root(og[g])
    for m in range(90):
        act = agent.play()
        stats = agent.get_stats()
        for g in range(G):
            a = oa[g].play(sims)
            assert a == act[g], (m, g)
            assert oa[g].stats().tobytes() == stats[g].tobytes(), (m, g)
            og[g].play(a)
            oa[g].update_root(og[g])
        game.play(act)
        agent.update_root(game)
        ended = game.end
        if ended.any():
            game.reset("ended")
            agent.update_root(game)
            for g in np.nonzero(ended)[0]:
                og[g].reset()
                oa[g].update_root(og[g])
    assert agent.store.counter("N_GC") == sum(o.n_gc for o in oa) >= 1


def test_play_cli_runs_a_short_game(capsys, tmp_path, monkeypatch):
    """play.main() itself, resolving the agent and the environment through the reference's own import lines
    (`from pyTetris import Tetris`, `import_module('agents.' + name)`, play.py:1,81-82): a few moves of ValueSimLP and of
    VanillaC, with the per-episode log line the dashboards parse when a game ends."""
    import play
    from tetris_mcts_amd import model as M
    monkeypatch.setattr(M, "EXP_PATH", str(tmp_path) + "/")
    monkeypatch.chdir(tmp_path)
    play.main(["--agent_type", "ValueSimLP", "--mcts_sims", "20", "--max_moves", "6", "--print_board_to_file"])
    assert os.path.getsize(tmp_path / "board_output") == 200
    play.main(["--agent_type", "VanillaC", "--mcts_sims", "8", "--endless", "--max_moves", "400", "--n_games", "3"])
    out = capsys.readouterr().out
    assert "Episode:" in out and "Lines Cleared:" in out
    # the distributional agent under both of its module names (the reference's file is DistValueSimOnline.py, its class DistValueSim)
    play.main(["--agent_type", "DistValueSim", "--mcts_sims", "30", "--max_moves", "3", "--n_games", "4"])
    play.main(["--agent_type", "DistValueSimOnline", "--mcts_sims", "30", "--max_moves", "2"])


# ---- TreeAgent's single calls (agent.cpp:825-833): update_root / expand(game) / new_node(game) / remove_nodes ----
def _tree_agent(G, max_nodes, seed):
    from tetris_mcts_amd import agents
    from tetris_mcts_amd.pyTetris import Tetris
    env_args = ((20, 10), 1, 0, 0)
    game = Tetris(*env_args, seed=seed, n_games=G)
    agent = agents.ValueSimC(sims=1, env=Tetris, env_args=env_args, n_games=G, max_nodes=max_nodes,
                             evaluator=hash_eval_torch)
    return game, agent, (lambda: Tetris(*env_args, seed=0, n_games=G))


@pytest.mark.parametrize("idx", [0, 1])
def test_tree_single_calls_vs_reference_treeagent(golden_dir, idx):
    """tests/golden/ref_treeagent.json: every index the reference's compiled TreeAgent returned from new_node(game) while
    being driven through update_root / expand(game), across two collections at a 1000-node pool (run 1)."""
    from test_oracle_agent import replay_treeagent, treeagent_script
    with open(os.path.join(golden_dir, "ref_treeagent.json")) as f:
        r = json.load(f)[idx]
    game, agent, mk = _tree_agent(1, r["max_nodes"], r["seed"])

    def check(i, kids, gk, root, ended):
        assert [kids, gk, root, int(ended)] == r["steps"][i], (i, kids, gk, root, ended, r["steps"][i])
    replay_treeagent(agent, game, mk, treeagent_script(len(r["steps"]), r["script_seed"]), check)
    assert not bool(agent.store.errors().any().item())
    assert agent.store.counter("N_GC") == (2 if idx == 1 else 0)


def test_tree_single_calls_batch_vs_oracle(oracle):
    """Four trees driven at once through the single calls, pool 1200: indices, then the whole reachable tree, against
    one oracle agent per game; remove_nodes() as a call of its own in the middle."""
    from test_oracle_agent import replay_treeagent, treeagent_script
    G, N, seed, steps = 4, 1200, 300, 200
    script = treeagent_script(steps, 11)
    game, agent, mk = _tree_agent(G, N, seed)
    got = []

    def rec(i, kids, gk, root, ended):
        got.append((np.stack(kids, 1).tolist(), np.stack(gk, 1).tolist(), np.asarray(root).tolist(),
                    np.asarray(ended).astype(int).tolist()))
        if i == 60:
            agent.remove_nodes()
    replay_treeagent(agent, game, mk, script, rec)
    assert not bool(agent.store.errors().any().item())
    gcs = 0
    for g in range(G):
        oa, og = oracle.Agent(3, max_nodes=N), oracle.Game(seed=seed + g)

        def check(i, kids, gk, root, ended):
            assert (kids, gk, root, int(ended)) == (got[i][0][g], got[i][1][g], got[i][2][g], got[i][3][g]), (g, i)
            if i == 60:
                oa.remove_nodes()
        replay_treeagent(oa, og, lambda: oracle.Game(seed=0), script, check)
        gcs += oa.n_gc
        dev, ref = agent.store.export_game(g), oa.arrays()
        mark = np.zeros(N, np.uint8)
        oracle.lib().orc_get_all_childs(oa.root, oracle.ptr(ref["child"]), N, oracle.ptr(mark))
        occ = np.nonzero(mark)[0]
        assert agent.store.t["gs"][g, 0].item() == oa.root
        for k in ("child", "score", "n_to_o"):
            assert np.array_equal(dev[k][occ], ref[k][occ]), (k, g)
        oo = np.unique(ref["n_to_o"][occ])
        assert dev["end_obs"][oo].tobytes() == ref["end_obs"][oo].tobytes()
    assert gcs >= 2 * G and agent.store.counter("N_GC") == gcs
