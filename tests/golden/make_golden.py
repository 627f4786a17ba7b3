"""Generate tests/golden/*.json|npz from the REFERENCE ITSELF (run in the build container only).

    python tests/golden/make_golden.py

* ref_uct_*.npz      : seeded random DAGs pushed through the reference's own core.cpp (compiled in place
                       into oracle/_ref/) -> expected traces / post-backup arrays / reachable sets.
* ref_agent_*.json   : the reference's own Python agents (agents/ValueSim.py, ValueSimLP.py imported from
                       /root/reference) playing on the oracle engine -> per-move (action, score, lines, stats).
* ref_valuenet.npz   : the reference's own Net (model/model_vv.py) on CPU fp32 -> weights, inputs, outputs.
* ref_norm_quantile.npy : special.h norm_quantile through the compiled reference (via policy_clt probing is
                       not possible; the table is pinned through select traces instead) - see tests.
The fixtures travel to the GPU box; /root/reference does not.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.dirname(os.path.abspath(__file__))

from oracle import binding as B  # noqa: E402
from oracle import ref_shims  # noqa: E402


def random_dag(rng, n_nodes, n_obs, expand_frac=0.6, low_visits=True):
    """A layered DAG in the reference's array layout (agents/agent.py:58-88)."""
    child = np.zeros((n_nodes, 7), np.int32)
    order = np.arange(1, n_nodes)
    for i in order:
        if i < n_nodes - 8 and rng.random() < expand_frac:
            lo = i + 1
            hi = min(n_nodes, i + 40)
            cs = rng.integers(lo, hi, 7)
            cs[rng.random(7) < 0.15] = 0
            if rng.random() < 0.3:
                cs[rng.integers(0, 7)] = cs[rng.integers(0, 7)]
            child[i] = cs
    n_to_o = rng.integers(1, n_obs, n_nodes).astype(np.int32)
    n_to_o[0] = 0
    score = np.round(rng.random(n_nodes) * 50).astype(np.float32)
    score[rng.random(n_nodes) < 0.3] = 7.0  # force score ties between duplicate observations
    visit = rng.integers(0 if low_visits else 1, 30, n_obs).astype(np.int32)
    value = (rng.standard_normal(n_obs) * 20).astype(np.float32)
    variance = (rng.random(n_obs) * 300).astype(np.float32)
    variance[rng.random(n_obs) < 0.1] = 0
    return child, n_to_o, score, visit, value, variance


def gen_uct():
    ref_shims.install()
    core = sys.modules["agents.cppmodule.core"]
    rng = np.random.default_rng(20260925)
    cases = {}
    n_cases = 24
    for ci in range(n_cases):
        n_nodes, n_obs = 400, int(rng.integers(60, 300))
        child, n_to_o, score, visit, value, variance = random_dag(rng, n_nodes, n_obs, low_visits=(ci % 2 == 0))
        if ci == 3:  # the n == 1 NaN case: a single child observation visited once (special.h t=1)
            child[1] = 0
            child[1][2] = 5
            visit[n_to_o[5]] = 1
        low = [1, 5][ci % 3 == 2]
        ref_shims.srand(1)
        traces = []
        v, val, var = visit.copy(), value.copy(), variance.copy()
        uniq = []
        for rep in range(12):
            root = int(rng.integers(1, 60))
            t = core.select_trace_obs(root, child, v, val, var, score, n_to_o, low)
            traces.append(np.asarray(t, np.int32))
            cn, co = core.get_unique_child_obs(root, child, score, n_to_o)
            uniq.append((root, list(cn), list(co)))
            mode = rep % 3
            if mode == 0:
                core.backup_trace_obs(t, v, val, var, n_to_o, score, float(rng.random() * 90), float(rng.random() * 500),
                                      0.999)
            else:
                leaf = int(t[-1])
                cn, co = core.get_unique_child_obs(leaf, child, score, n_to_o)
                k = len(cn)
                end = np.zeros(n_nodes, bool)
                if mode == 2 and k:
                    end[cn[0]] = True
                _v = (rng.random(k) * 60).astype(np.float32)
                _var = (rng.random(k) * 400).astype(np.float32)
                core.backup_trace_obs_LP(t, v, val, var, n_to_o, score, end, list(cn), list(co), _v, _var, 0.999, False,
                                         True)
        reach = np.array(sorted(core.get_all_childs(int(rng.integers(1, 60)), child)), np.int32)
        cases["c%d" % ci] = dict(child=child, n_to_o=n_to_o, score=score, visit0=visit, value0=value,
                                 variance0=variance, low=low, traces=traces, uniq=uniq, visit1=v, value1=val,
                                 variance1=var, reach=reach)
    # flatten to npz: the generating seed is fixed, tests regenerate the inputs and only need the outputs
    flat = {}
    for name, c in cases.items():
        flat[name + "_tracecat"] = np.concatenate(c["traces"])
        flat[name + "_tracelen"] = np.array([len(t) for t in c["traces"]], np.int32)
        flat[name + "_uniq"] = np.array([[r] + cn + [0] * (7 - len(cn)) + co + [0] * (7 - len(co)) + [len(cn)]
                                         for r, cn, co in c["uniq"]], np.int32)
        for k in ("visit1", "value1", "variance1", "reach"):
            flat[name + "_" + k] = c[k]
    np.savez_compressed(os.path.join(OUT, "ref_uct.npz"), **flat)
    print("ref_uct.npz:", len(cases), "cases")


def gen_vanilla():
    """The reference's Vanilla agent (agents/Vanilla.py, random rollouts with random.randint) - BASELINE configs[0]."""
    import random
    ref_shims.install()
    from pyTetris import Tetris
    out = []
    for sims, max_nodes, seed, rseed, moves in ((100, 60000, 41, 0, 40), (30, 4000, 42, 7, 120)):
        ref_shims.srand(1)
        random.seed(rseed)
        agent = ref_shims.make_agent("Vanilla", sims, max_nodes=max_nodes)
        game = Tetris((20, 10), 1, 0, 0, seed)
        agent.update_root(game)
        rec = []
        for _ in range(moves):
            a = int(agent.play())
            stats = agent.get_stats()
            game.play(a)
            agent.update_root(game)
            rec.append([a, int(game.score), int(game.line_clears), stats.astype("<f4").tobytes().hex()])
            if game.end:
                game.reset()
                agent.update_root(game)
        out.append(dict(sims=sims, max_nodes=max_nodes, seed=seed, random_seed=rseed, moves=rec))
        print("Vanilla sims %d moves %d final score %d" % (sims, len(rec), rec[-1][1]))
    with open(os.path.join(OUT, "ref_vanilla.json"), "w") as f:
        json.dump(out, f)


def gen_uct_mixture():
    """backup_trace_obs_LP with mixture / non-averaged flags (core.h:262-301,303-381) through the reference's core.cpp."""
    ref_shims.install()
    core = sys.modules["agents.cppmodule.core"]
    rng = np.random.default_rng(77)
    flat = {}
    for ci in range(8):
        child, n_to_o, score, visit, value, variance = random_dag(rng, 300, 150, low_visits=False)
        ref_shims.srand(1)
        v, val, var = visit.copy(), value.copy(), variance.copy()
        for rep in range(10):
            # a trace that ends in an EXPANDED node, so the LP child list is non-empty
            root = int(rng.integers(1, 40))
            t = np.asarray(core.select_trace_obs(root, child, v, val, var, score, n_to_o, 1), np.int32)
            if len(t) > 1:
                t = t[:-1].copy()
            leaf = int(t[-1])
            cn, co = core.get_unique_child_obs(leaf, child, score, n_to_o)
            k = len(cn)
            end = np.zeros(300, bool)
            _v = (rng.random(k) * 60).astype(np.float32)
            _var = (rng.random(k) * 400).astype(np.float32)
            mixture, averaged = bool(rep & 1), bool(rep & 2)
            core.backup_trace_obs_LP(t, v, val, var, n_to_o, score, end, list(cn), list(co), _v, _var, 0.98, mixture, averaged)
        flat["m%d_visit1" % ci], flat["m%d_value1" % ci], flat["m%d_variance1" % ci] = v, val, var
    np.savez_compressed(os.path.join(OUT, "ref_uct_mixture.npz"), **flat)
    print("ref_uct_mixture.npz: 8 cases")


def play_ref(name, sims, max_nodes, seed, max_moves, evaluator):
    ref_shims.install()
    from pyTetris import Tetris
    ref_shims.srand(1)
    agent = ref_shims.make_agent(name, sims, max_nodes=max_nodes, evaluator=evaluator)
    game = Tetris((20, 10), 1, 0, 0, seed)
    agent.update_root(game)
    moves = []
    while len(moves) < max_moves:
        a = int(agent.play())
        stats = agent.get_stats()
        game.play(a)
        agent.update_root(game)
        moves.append([a, int(game.score), int(game.line_clears), stats.astype("<f4").tobytes().hex()])
        if game.end:
            game.reset()
            agent.update_root(game)
    return moves


def hash_eval(states):
    k = states.shape[0]
    s = np.ascontiguousarray(states, np.int8)
    v = np.zeros(k, np.float32)
    var = np.zeros(k, np.float32)
    B.lib().orc_hash_eval(None, B.ptr(s), k, B.ptr(v), B.ptr(var))
    return v, var


def gen_agents(params):
    def net_eval(states):
        k = states.shape[0]
        s = np.ascontiguousarray(states, np.int8)
        v = np.zeros(k, np.float32)
        var = np.zeros(k, np.float32)
        B.lib().orc_valuenet_forward(B.ptr(params), B.ptr(s), k, B.ptr(v), B.ptr(var))
        return v, var
    runs = [
        dict(name="ValueSim", sims=40, max_nodes=8000, seed=11, max_moves=200, evaluator="hash"),
        dict(name="ValueSimLP", sims=30, max_nodes=8000, seed=12, max_moves=200, evaluator="hash"),
        dict(name="ValueSim", sims=25, max_nodes=100000, seed=13, max_moves=30, evaluator="valuenet"),
        dict(name="ValueSimLP", sims=12, max_nodes=100000, seed=14, max_moves=24, evaluator="valuenet"),
    ]
    out = []
    for r in runs:
        ev = hash_eval if r["evaluator"] == "hash" else net_eval
        moves = play_ref(r["name"], r["sims"], r["max_nodes"], r["seed"], r["max_moves"], ev)
        out.append(dict(r, moves=moves))
        print(r["name"], r["evaluator"], "moves", len(moves), "final score", moves[-1][1], "lines", moves[-1][2])
    with open(os.path.join(OUT, "ref_agents.json"), "w") as f:
        json.dump(out, f)


def gen_valuenet():
    """Weights + outputs of the reference's own Net (model/model_vv.py:13-52) on CPU fp32."""
    ref_shims.install()
    import torch
    from model.model_vv import Net
    torch.manual_seed(0)
    torch.set_num_threads(1)
    net = Net().eval()
    sd = net.state_dict()
    order = ["head.conv1.weight", "head.conv1.bias", "head.conv2.weight", "head.conv2.bias", "head.conv3.weight",
             "head.conv3.bias", "head.fc1.weight", "head.fc1.bias", "head.fc_out.weight", "head.fc_out.bias",
             "out_ubound", "out_lbound"]
    assert sorted(sd.keys()) == sorted(order), list(sd.keys())
    params = np.concatenate([sd[k].detach().numpy().ravel() for k in order]).astype(np.float32)
    assert params.size == 478342
    # a second, "trained-like" parameter set with larger activations (stress for the 1e-4 tolerance)
    rng = np.random.default_rng(7)
    params2 = params.copy()
    params2[:478338] *= 1.6
    params2[478338:478340] = [250.0, 4000.0]
    # inputs: boards in the style of tools/test.py:23-28 (random cells below a cleared top, one falling piece)
    states = np.zeros((64, 20, 10), np.int8)
    for i in range(64):
        h = int(rng.integers(0, 16))
        states[i, 20 - h:, :] = (rng.random((h, 10)) < 0.7).astype(np.int8)
        states[i, 1:3, 5:7] = -1 if i % 4 else 0
    outs = []
    for p in (params, params2):
        off = 0
        with torch.no_grad():
            for k in order:
                n = sd[k].numel()
                sd[k].copy_(torch.from_numpy(p[off:off + n].reshape(sd[k].shape)))
                off += n
            y = net(torch.from_numpy(states.astype(np.float32))[:, None]).numpy()
        outs.append(y.astype(np.float32))
    np.savez_compressed(os.path.join(OUT, "ref_valuenet.npz"), params=params, params2=params2, states=states,
                        out=outs[0], out2=outs[1])
    print("ref_valuenet.npz: out range", outs[0].min(0), outs[0].max(0), outs[1].min(0), outs[1].max(0))
    return params


def gen_cppagent():
    """The reference's all-C++ agent (agents/cppmodule/agent.cpp MCTSAgent, compiled in place into oracle/_ref/) on the
    oracle engine with the hash evaluator: per-move (action, score, lines) for LP and non-LP."""
    ref_shims.install()
    from pyTetris import Tetris
    agent_mod = sys.modules["agents.cppmodule.agent"]

    def ev_lp(obs):
        a = np.asarray(obs).astype(np.int8)
        v, var = hash_eval(a.reshape(a.shape[0], 20, 10))
        return [v.tolist(), var.tolist()]

    def ev_single(obs):
        a = np.asarray(obs).astype(np.int8)
        v, var = hash_eval(a.reshape(1, 20, 10))
        return [float(v[0]), float(var[0])]
    out = []
    for lp, sims, max_nodes, seed, moves in ((True, 30, 8000, 31, 200), (False, 40, 8000, 32, 200)):
        ref_shims.srand(1)
        agent = agent_mod.MCTSAgent(sims, max_nodes, True, 0.999, False, ev_lp if lp else ev_single, 0, lp)
        game = Tetris((20, 10), 1, 0, 0, seed)
        agent.update_root(game)
        rec = []
        for _ in range(moves):
            a = int(agent.play())
            game.play(a)
            agent.update_root(game)
            rec.append([a, int(game.score), int(game.line_clears)])
            if game.end:
                game.reset()
                agent.update_root(game)
        out.append(dict(lp=lp, sims=sims, max_nodes=max_nodes, seed=seed, moves=rec))
        print("MCTSAgent LP=%s moves %d final score %d" % (lp, len(rec), rec[-1][1]))
    with open(os.path.join(OUT, "ref_cppagent.json"), "w") as f:
        json.dump(out, f)


def gen_online():
    """The reference's OnlineMCTSAgent (agents/cppmodule/agent.cpp:569-816, compiled in place) with online=True:
    every train() payload (memory contents at the moment the accumulation policy fires), for the four policies.
    Small pool -> frequent GC; small memory -> policies 0/1 trim, policy 2 fires on a full memory."""
    import hashlib
    ref_shims.install()
    from pyTetris import Tetris
    agent_mod = sys.modules["agents.cppmodule.agent"]

    def ev_lp(obs):
        a = np.asarray(obs).astype(np.int8)
        v, var = hash_eval(a.reshape(a.shape[0], 20, 10))
        return [v.tolist(), var.tolist()]
    out = []
    cfgs = [dict(policy=0, memory_size=400, episodes_per_train=3, growth=100),
            dict(policy=1, memory_size=400, episodes_per_train=3, growth=100),
            dict(policy=2, memory_size=600, episodes_per_train=3, growth=100),
            dict(policy=3, memory_size=2000, episodes_per_train=3, growth=300)]
    for c in cfgs:
        sims, max_nodes, min_visit, seed, moves = 30, 8000, 3, 41 + c["policy"], 900
        calls, state = [], dict(move=0)

        def train(m_state, m_value, m_variance, m_visit, size, calls=calls, state=state):
            st = np.ascontiguousarray(np.asarray(m_state)[:size]).astype(np.int8)
            val = np.ascontiguousarray(np.asarray(m_value)[:size], "<f4").reshape(-1)
            var = np.ascontiguousarray(np.asarray(m_variance)[:size], "<f4").reshape(-1)
            vis = np.ascontiguousarray(np.asarray(m_visit)[:size], "<f4").reshape(-1)
            calls.append(dict(move=state["move"], size=int(size), states_sha1=hashlib.sha1(st.tobytes()).hexdigest(),
                              value=val.tobytes().hex(), variance=var.tobytes().hex(), visit=vis.tobytes().hex()))
        ref_shims.srand(1)
        # agent.cpp:29's std::mt19937 mt(123) is process-wide and only policy 0 draws from it: policy 0 runs first
        agent = agent_mod.OnlineMCTSAgent(sims, max_nodes, True, c["policy"], c["memory_size"], c["episodes_per_train"],
                                          c["growth"], min_visit, True, 0.999, False, ev_lp, 0, train, True)
        game = Tetris((20, 10), 1, 0, 0, seed)
        agent.update_root(game)
        rec, episodes = [], 0
        for m in range(moves):
            state["move"] = m
            a = int(agent.play())
            game.play(a)
            agent.update_root(game)
            rec.append(a)
            if game.end:
                episodes += 1
                game.reset()
                agent.update_root(game)
        out.append(dict(c, sims=sims, max_nodes=max_nodes, min_visit=min_visit, seed=seed, actions=rec, train_calls=calls,
                        episodes=episodes))
        print("OnlineMCTSAgent policy %d: %d moves, %d episodes, %d train calls, sizes %s" % (
            c["policy"], moves, episodes, len(calls), [k["size"] for k in calls][:12]), file=sys.stderr)
    with open(os.path.join(OUT, "ref_online_cpp.json"), "w") as f:
        json.dump(out, f)


def gen_online_py():
    """The reference's Python ValueSim / ValueSimLP with online=True (ValueSim.py:101-159): the replay tuples
    store_nodes() has put into agent.memory at every pool-exhaustion GC (train_nodes is replaced by a recorder that
    empties the memory the way a training pass does, ValueSim.py:182)."""
    import hashlib
    out = []
    for name, sims, seed in (("ValueSim", 40, 51), ("ValueSimLP", 30, 52)):
        ref_shims.install()
        from pyTetris import Tetris
        ref_shims.srand(1)
        agent = ref_shims.make_agent(name, sims, max_nodes=8000, evaluator=hash_eval, online=True, memory_size=50000)
        agent.min_visits_to_store = 3       # ValueSimLP hard-codes 25 in its constructor (ValueSimLP.py:11)
        calls, state = [], dict(move=0)

        def record(dump_data=True, agent=agent, calls=calls, state=state):
            n = agent.memory_index
            st, val, var, w = [m[:n] for m in agent.memory]
            calls.append(dict(move=state["move"], size=int(n),
                              states_sha1=hashlib.sha1(np.ascontiguousarray(st).astype(np.int8).tobytes()).hexdigest(),
                              value=np.ascontiguousarray(val, "<f4").tobytes().hex(),
                              variance=np.ascontiguousarray(var, "<f4").tobytes().hex(),
                              visit=np.ascontiguousarray(w, "<f4").tobytes().hex()))
            agent.memory_index = 0
        agent.train_nodes = record
        game = Tetris((20, 10), 1, 0, 0, seed)
        agent.update_root(game)
        acts = []
        for m in range(600):
            state["move"] = m
            a = int(agent.play())
            game.play(a)
            agent.update_root(game)
            acts.append(a)
            if game.end:
                game.reset()
                agent.update_root(game)
        out.append(dict(name=name, sims=sims, seed=seed, max_nodes=8000, min_visits_to_store=3, actions=acts, harvests=calls))
        print(name, "online: GCs", len(calls), "sizes", [c["size"] for c in calls], file=sys.stderr)
    with open(os.path.join(OUT, "ref_online_py.json"), "w") as f:
        json.dump(out, f)


def gen_agents_env():
    """Reference agents under non-default environments: app > 1 (gravity every `app` actions: a trace can then contain
    the same observation twice, the case the parallel backup must serialise), guideline scoring off / uniform
    randomizer; Python ValueSim / ValueSimLP and the compiled C++ MCTSAgent.  Hash evaluator, small pools (GC)."""
    ref_shims.install()
    from pyTetris import Tetris
    agent_mod = sys.modules["agents.cppmodule.agent"]

    def ev_lp(obs):
        a = np.asarray(obs).astype(np.int8)
        v, var = hash_eval(a.reshape(a.shape[0], 20, 10))
        return [v.tolist(), var.tolist()]
    runs = [dict(name="ValueSim", sims=30, max_nodes=8000, seed=61, max_moves=120, env=[2, 1, 1]),
            dict(name="ValueSimLP", sims=25, max_nodes=8000, seed=62, max_moves=120, env=[3, 0, 0]),
            dict(name="MCTSAgent", sims=30, max_nodes=8000, seed=63, max_moves=160, env=[2, 0, 1])]
    out = []
    for r in runs:
        app, scoring, randomizer = r["env"]
        ref_shims.srand(1)
        if r["name"] == "MCTSAgent":
            agent = agent_mod.MCTSAgent(r["sims"], r["max_nodes"], True, 0.999, False, ev_lp, 0, True)
        else:
            agent = ref_shims.make_agent(r["name"], r["sims"], max_nodes=r["max_nodes"], evaluator=hash_eval,
                                         env_args=((20, 10), app, scoring, randomizer))
        game = Tetris((20, 10), app, scoring, randomizer, r["seed"])
        agent.update_root(game)
        moves = []
        while len(moves) < r["max_moves"]:
            a = int(agent.play())
            stats = agent.get_stats().astype("<f4").tobytes().hex() if hasattr(agent, "get_stats") else ""
            game.play(a)
            agent.update_root(game)
            moves.append([a, int(game.score), int(game.line_clears), stats])
            if game.end:
                game.reset()
                agent.update_root(game)
        out.append(dict(r, moves=moves))
        print(r["name"], r["env"], "moves", len(moves), "final score", moves[-1][1], "lines", moves[-1][2], file=sys.stderr)
    with open(os.path.join(OUT, "ref_agents_env.json"), "w") as f:
        json.dump(out, f)


def gen_vanillac():
    """The reference's VanillaC (agents/VanillaC.py: the compiled MCTSAgent with evaluator type 1 = a Python random playout
    of a copy of the leaf, `game.play(randint(0, 7))`, variance 1e5, gamma 0.99) on the oracle engine."""
    import random
    ref_shims.install()
    from pyTetris import Tetris
    agent_mod = sys.modules["agents.cppmodule.agent"]

    def random_playout(game):               # agents/VanillaC.py:5-8, verbatim semantics
        while not game.end:
            game.play(random.randint(0, 7))
        return game.score, 1e5
    out = []
    for sims, max_nodes, seed, rseed, moves in ((100, 60000, 51, 0, 40), (40, 40000, 53, 9, 120), (30, 3000, 52, 5, 150)):
        ref_shims.srand(1)
        random.seed(rseed)
        agent = agent_mod.MCTSAgent(sims, max_nodes, True, 0.99, False, random_playout, 1, False)
        game = Tetris((20, 10), 1, 0, 0, seed)
        agent.update_root(game)
        rec = []
        for _ in range(moves):
            a = int(agent.play())
            game.play(a)
            agent.update_root(game)
            rec.append([a, int(game.score), int(game.line_clears)])
            if game.end:
                game.reset()
                agent.update_root(game)
        out.append(dict(sims=sims, max_nodes=max_nodes, seed=seed, random_seed=rseed, moves=rec))
        print("VanillaC sims %d moves %d final score %d lines %d" % (sims, len(rec), rec[-1][1], rec[-1][2]))
    with open(os.path.join(OUT, "ref_vanillac.json"), "w") as f:
        json.dump(out, f)


def treeagent_script(steps, script_seed):
    """(action expanded below the root, action played) per step: the script both the reference and the replays follow."""
    rs = np.random.RandomState(script_seed)
    return [[int(rs.randint(7)), int(rs.randint(7))] for _ in range(steps)]


def gen_treeagent():
    """ref_treeagent.json: the reference's compiled TreeAgent (agents/cppmodule/agent.cpp:84-413, exported at :825-833)
    driven through its single calls - update_root, expand(game), new_node(game) - on the oracle engine.  Per step: expand
    the root, ask new_node for its seven successors (existing nodes: their indices), expand one successor, ask new_node
    for ITS successors, play a scripted action, update_root, new_node(root game).  The index sequences pin new_node /
    expand / the collection at an exhausted pool (run 2: pool of 1000) as the pybind module itself answers them."""
    ref_shims.install()
    from pyTetris import Tetris
    agent_mod = sys.modules["agents.cppmodule.agent"]
    out = []
    for max_nodes, seed, steps, sseed in ((6000, 81, 80, 5), (1000, 82, 260, 6)):
        agent = agent_mod.TreeAgent(max_nodes, True, 0)
        game = Tetris((20, 10), 1, 0, 0, seed)
        tmp, tmp2 = Tetris((20, 10), 1, 0, 0, 0), Tetris((20, 10), 1, 0, 0, 0)
        agent.update_root(game)
        rec = []
        for a1, a in treeagent_script(steps, sseed):
            agent.expand(game)
            kids = []
            for b in range(7):
                tmp.copy_from(game)
                tmp.play(b)
                kids.append(int(agent.new_node(tmp)))
            tmp.copy_from(game)
            tmp.play(a1)
            agent.expand(tmp)
            gk = []
            for b in range(7):
                tmp2.copy_from(tmp)
                tmp2.play(b)
                gk.append(int(agent.new_node(tmp2)))
            game.play(a)
            agent.update_root(game)
            root = int(agent.new_node(game))
            ended = bool(game.end)
            if ended:
                game.reset()
                agent.update_root(game)
            rec.append([kids, gk, root, int(ended)])
        out.append(dict(max_nodes=max_nodes, seed=seed, script_seed=sseed, steps=rec))
        print("TreeAgent pool %d: %d steps, highest index %d, %d episode ends" % (
            max_nodes, len(rec), max(max(r[0] + r[1]) for r in rec), sum(r[3] for r in rec)))
    with open(os.path.join(OUT, "ref_treeagent.json"), "w") as f:
        json.dump(out, f)


def gen_distpy():
    """ref_distpy.npz: the reference's numba kernels of agents/core_distributional.py (shift_distribution :12-37,
    policy_dist :66-79, backup_trace_distributional :108-124) run as plain Python (numba shimmed away; they are `fastmath`
    numba code, so their compiled bit patterns are not defined - the fixtures are held to a float tolerance).  policy_dist
    cases are kept only when the winning bound leads by a clear margin."""
    ref_shims.install()
    from agents import core_distributional as cd
    from agents.special import norm_quantile
    rng = np.random.default_rng(77)
    out = {}
    # shift_distribution: shifts >= 0 (the only ones backup produces: a leaf's score is never below an ancestor's)
    n = 0
    for bins, vmax in ((50, 5000.0), (31, 310.0), (64, 100.0)):
        for x in (0.0, 1.0, 99.5, 100.0, 250.25, 1234.0, vmax - 1.0, vmax + 500.0):
            d = rng.random(bins).astype(np.float32)
            d /= d.sum()
            r = cd.shift_distribution(d, x, 0.0, vmax)
            out.update({"s_dist_%d" % n: d, "s_x_%d" % n: x, "s_vmax_%d" % n: vmax, "s_out_%d" % n: np.asarray(r, np.float32)})
            n += 1
    out["s_n"] = n
    # policy_dist
    n = 0
    while n < 40:
        n_nodes = 30
        ns = np.zeros((n_nodes, 5), np.float32)
        ns[:, 0] = rng.integers(1, 60, n_nodes)
        ns[:, 1] = rng.random(n_nodes) * 400
        ns[:, 2] = rng.integers(0, 30, n_nodes) * 10
        ns[:, 3] = rng.random(n_nodes) * 2000
        nd = rng.random((n_nodes, 50)).astype(np.float32)
        nc = int(rng.integers(1, 8))
        cn = [int(c) for c in rng.choice(np.arange(1, n_nodes), nc, replace=False)]
        cur = float(ns[0, 2])
        got = int(cd.policy_dist(cn, ns, nd, cur, 0.0, 5000.0))
        tot = float(sum(float(ns[c, 0]) for c in cn))
        q = sorted(float(ns[c, 1]) + float(ns[c, 2]) - cur + norm_quantile(tot) * np.sqrt(float(ns[c, 3]) / (float(ns[c, 0]) + 1e-3))
                   for c in cn)
        if len(q) > 1 and q[-1] - q[-2] < 1e-2:
            continue
        out.update({"p_stats_%d" % n: ns, "p_nodes_%d" % n: np.asarray(cn, np.int32), "p_cur_%d" % n: cur, "p_out_%d" % n: got})
        n += 1
    out["p_n"] = n
    # backup_trace_distributional
    n = 0
    for bins, vmax, tl in ((50, 5000.0, 9), (50, 5000.0, 1), (32, 640.0, 20), (64, 2000.0, 14)):
        for rep in range(3):
            n_nodes = 40
            ns = np.zeros((n_nodes, 5), np.float32)
            ns[:, 0] = rng.integers(0, 25, n_nodes)
            ns[:, 1] = rng.random(n_nodes) * vmax * 0.3
            ns[:, 2] = rng.integers(0, 20, n_nodes) * 10
            ns[:, 4] = rng.random(n_nodes) * 3000 * (ns[:, 0] > 0)
            ns[:, 3] = np.where(ns[:, 0] > 1, ns[:, 4] / np.maximum(ns[:, 0] - 1, 1), 0)
            nd = rng.random((n_nodes, bins)).astype(np.float32)
            nd /= nd.sum(1, keepdims=True)
            nd[ns[:, 0] == 0] = 0
            trace = rng.choice(np.arange(1, n_nodes), tl, replace=False).astype(np.int32)
            d = rng.random(bins).astype(np.float32)
            d /= d.sum()
            r = float(ns[trace, 2].max() + 10 * rng.integers(0, 8))
            ns_in, nd_in = ns.copy(), nd.copy()
            cd.backup_trace_distributional(trace, ns, nd, r, d, 0.0, vmax)
            out.update({"b_stats_in_%d" % n: ns_in, "b_dist_in_%d" % n: nd_in, "b_trace_%d" % n: trace, "b_r_%d" % n: r,
                        "b_leaf_%d" % n: d, "b_vmax_%d" % n: vmax, "b_stats_out_%d" % n: ns, "b_dist_out_%d" % n: nd})
            n += 1
    out["b_n"] = n
    np.savez_compressed(os.path.join(OUT, "ref_distpy.npz"), **out)
    print("ref_distpy.npz: %d shift, %d policy, %d backup cases" % (out["s_n"], out["p_n"], out["b_n"]))


def gen_dist():
    """ref_dist.npz: the reference's distribution helpers (core.h:387-449, compiled through oracle/ref_dist_shim.cpp) on
    seeded categorical distributions.  Only cases in which the reference's write to result[bins] adds exactly 0 or does not
    happen (it is a heap overrun otherwise - `python -c` with shift 3.0 aborts with "corrupted size vs. prev_size")."""
    B.build(ref=True)
    m = B.load_ref_native("dist")
    rng = np.random.default_rng(77)
    cases = []
    for bins in (50, 50, 32, 64):
        for vmin, vmax in ((0.0, 100.0), (-20.0, 380.0)):
            for shift, scale in ((0.0, 1.0), (-7.3, 1.0), (-30.0, 0.99), (-0.5, 0.9), (-55.0, 0.999), (0.0, 0.95)):
                d = rng.random(bins).astype(np.float32)
                d[rng.random(bins) < 0.2] = 0
                d /= d.sum()
                delta = (vmax - vmin) / bins
                lb = np.maximum(np.arange(bins) * scale + shift / delta, 0.0)
                ub = np.minimum(lb + scale, float(bins))
                frac = np.floor(ub) - lb
                assert ((np.floor(ub) < bins) | (frac == 1.0)).all()      # no harmful overrun in the reference
                out = np.asarray(m.transform_distribution(d, vmin, vmax, shift, scale), np.float32)
                mv = m.mean_variance_dist(d, vmin, vmax)
                cases.append(dict(dist=d, vmin=vmin, vmax=vmax, shift=shift, scale=scale, out=out,
                                  mean=m.mean_dist(d, vmin, vmax), mv=np.asarray(mv, np.float64)))
    np.savez_compressed(os.path.join(OUT, "ref_dist.npz"), n=len(cases),
                        **{"%s_%d" % (k, i): np.asarray(c[k]) for i, c in enumerate(cases) for k in c})
    print("ref_dist.npz:", len(cases), "cases")


def gen_distnet():
    """ref_distnet.npz: the reference's distributional `Net` (model/model_distributional.py:18-57) on CPU fp32 under
    manual_seed(0): its state_dict, 16 inputs [16,1,22,10] in {-1,0,1}, softmax outputs and log_prob."""
    ref_shims.install()
    _legacy_torch_overloads()
    import torch
    torch.set_num_threads(1)
    from model.model_distributional import Net
    torch.manual_seed(0)
    net = Net(atoms=50).eval()
    rng = np.random.default_rng(3)
    x = np.zeros((16, 1, 22, 10), np.float32)
    for i in range(16):
        h = int(rng.integers(0, 16))
        x[i, 0, 22 - h:, :] = (rng.random((h, 10)) < 0.7)
        x[i, 0, 3:5, 4:6] = -1
    with torch.no_grad():
        y = net(torch.from_numpy(x)).numpy()
        lp = net.log_prob(torch.from_numpy(x)).numpy()
    sd = {k.replace(".", "__"): v.numpy() for k, v in net.state_dict().items()}
    np.savez_compressed(os.path.join(OUT, "ref_distnet.npz"), x=x, y=y, lp=lp, **sd)
    print("ref_distnet.npz: y sums", y.sum(1)[:3], "keys", sorted(sd))


def _legacy_torch_overloads():
    """model/yogi.py and model/model_vv.py call `add(number, tensor)`, `add_(number, tensor)`, `addcmul_(number, t, t)` and
    `addcdiv_(number, t, t)`, signatures PyTorch removed after 1.x.  Re-add them (number first = `alpha` / `value`) so the
    reference's training code runs unmodified."""
    import numbers
    import torch
    T = torch.Tensor
    if getattr(T, "_tm_legacy", False):
        return
    orig = {n: getattr(T, n) for n in ("add", "add_", "addcmul_", "addcdiv_")}

    def _num(x):
        return isinstance(x, numbers.Number)

    def add(self, *a, **k):
        return orig["add"](self, a[1], alpha=a[0]) if len(a) == 2 and _num(a[0]) else orig["add"](self, *a, **k)

    def add_(self, *a, **k):
        return orig["add_"](self, a[1], alpha=a[0]) if len(a) == 2 and _num(a[0]) else orig["add_"](self, *a, **k)

    def addcmul_(self, *a, **k):
        return orig["addcmul_"](self, a[1], a[2], value=a[0]) if len(a) == 3 and _num(a[0]) else orig["addcmul_"](self, *a, **k)

    def addcdiv_(self, *a, **k):
        return orig["addcdiv_"](self, a[1], a[2], value=a[0]) if len(a) == 3 and _num(a[0]) else orig["addcdiv_"](self, *a, **k)
    T.add, T.add_, T.addcmul_, T.addcdiv_, T._tm_legacy = add, add_, addcmul_, addcdiv_, True


def gen_training():
    """ref_training.npz: the reference's own optimiser, loss and training step on fixed tensors (CPU, 1 thread).
      * model/yogi.py:39-90 `Yogi.step`: 10 steps on a [6,5] parameter, fp64 and fp32, lr 1e-3 eps 1e-3 wd 1e-3
        (the settings of model_vv.py:132) -> the parameter after every step;
      * model/model_vv.py:94-101 `GaussianLL.forward` on 64 samples;
      * model/model_vv.py:136-150 `Model_VV._loss` + model/model.py:97-116 `Model.train` (zero_grad, loss, backward, step)
        on the reference's own Net under manual_seed(0): loss values, gradient norms and all 478342 parameters
        after 4 steps (weighted)."""
    ref_shims.install()
    _legacy_torch_overloads()
    import torch
    torch.set_num_threads(1)
    from model.yogi import Yogi
    from model.model_vv import GaussianLL, Model_VV
    rng = np.random.default_rng(11)
    out = {}
    p0 = rng.standard_normal((6, 5))
    grads = rng.standard_normal((10, 6, 5)) * np.array([1.0, 0.1, 3.0, 1e-3, 1.0, 10.0])[None, :, None]
    out["yogi_p0"], out["yogi_grads"] = p0, grads
    for dt, name in ((torch.float64, "f64"), (torch.float32, "f32")):
        p = torch.nn.Parameter(torch.tensor(p0, dtype=dt))
        opt = Yogi([p], lr=1e-3, eps=1e-3, weight_decay=1e-3)
        traj = []
        for g in grads:
            p.grad = torch.tensor(g, dtype=dt)
            opt.step()
            traj.append(p.detach().numpy().copy())
        out["yogi_traj_" + name] = np.stack(traj)
    vp = (rng.random((64, 1)) * 50 + 0.1).astype(np.float32)
    mp = (rng.standard_normal((64, 1)) * 30).astype(np.float32)
    var = (rng.random((64, 1)) * 40 + 0.1).astype(np.float32)
    mean = (rng.standard_normal((64, 1)) * 30).astype(np.float32)
    out["ll_in"] = np.stack([vp, mp, var, mean])
    out["ll_out"] = GaussianLL()(torch.from_numpy(vp.copy()), torch.from_numpy(mp.copy()), torch.from_numpy(var.copy()),
                                 torch.from_numpy(mean.copy())).numpy()
    torch.manual_seed(0)
    m = Model_VV(use_cuda=False)
    m.training(True)
    n = 48
    states = np.zeros((n, 1, 20, 10), np.float32)
    for i in range(n):
        h = int(rng.integers(0, 14))
        states[i, 0, 20 - h:, :] = (rng.random((h, 10)) < 0.7)
        states[i, 0, 1:3, 4:6] = -1
    values = (rng.random((n, 1)) * 60).astype(np.float32)
    variances = (rng.random((n, 1)) * 200).astype(np.float32)
    variances[:5] = 0.01                                   # below the clip (model_vv.py:141: clamp at 0.1)
    weights = rng.integers(10, 400, (n, 1)).astype(np.float32)
    weights = weights / weights.mean()
    out["tr_states"], out["tr_values"], out["tr_variances"], out["tr_weights"] = states, values, variances, weights
    order = ["head.conv1.weight", "head.conv1.bias", "head.conv2.weight", "head.conv2.bias", "head.conv3.weight",
             "head.conv3.bias", "head.fc1.weight", "head.fc1.bias", "head.fc_out.weight", "head.fc_out.bias",
             "out_ubound", "out_lbound"]
    flat = lambda: np.concatenate([m.model.state_dict()[k].detach().numpy().ravel() for k in order]).astype(np.float32)  # noqa: E731
    out["tr_params0"] = flat()
    losses, gnorms = [], []
    for it in range(4):
        r = m.train([states.copy(), values.copy(), variances.copy(), weights.copy()], weighted=True)
        losses.append(r["loss"])
        gnorms.append(r["grad_norm"])
    out["tr_losses"], out["tr_gnorms"], out["tr_params4"] = np.asarray(losses, np.float64), np.asarray(gnorms, np.float64), flat()
    val = m.compute_loss([states.copy(), values.copy(), variances.copy(), weights.copy()], weighted=True, chunksize=20)
    out["tr_val"] = np.asarray([val["loss"], val["loss_std"]], np.float64)
    # optimiser state layout of a reference checkpoint (model/model.py:152-160): 12 parameters in one group
    sd = m.optimizer.state_dict()
    out["opt_group_params"] = np.asarray(sd["param_groups"][0]["params"], np.int64)
    np.savez_compressed(os.path.join(OUT, "ref_training.npz"), **out)
    print("ref_training.npz: losses", losses, "grad norms", gnorms, "val", val)


if __name__ == "__main__":
    which = sys.argv[1:] or ["uct", "valuenet", "agents", "cppagent", "mixture", "vanilla", "online", "online_py", "agents_env", "training", "vanillac", "dist", "distnet", "treeagent", "distpy"]
    params = None
    if "uct" in which:
        gen_uct()
    if "valuenet" in which:
        params = gen_valuenet()
    if "cppagent" in which:
        gen_cppagent()
    if "mixture" in which:
        gen_uct_mixture()
    if "vanilla" in which:
        gen_vanilla()
    if "online" in which:
        gen_online()
    if "online_py" in which:
        gen_online_py()
    if "agents_env" in which:
        gen_agents_env()
    if "training" in which:
        gen_training()
    if "vanillac" in which:
        gen_vanillac()
    if "treeagent" in which:
        gen_treeagent()
    if "distpy" in which:
        gen_distpy()
    if "dist" in which:
        gen_dist()
    if "distnet" in which:
        gen_distnet()
    if "agents" in which:
        if params is None:
            params = np.load(os.path.join(OUT, "ref_valuenet.npz"))["params"]
        gen_agents(params)
