"""ORACLE — test infrastructure only.  ctypes bindings for oracle/_build/liboracle.so."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = os.path.join(HERE, "_build")
REFOUT = os.path.join(HERE, "_ref")

GAME_DTYPE = np.dtype([
    ("rows", "<u2", (20,)), ("piece", "u1"), ("rot", "u1"), ("x", "i1"), ("y", "i1"),
    ("drop_ctr", "u1"), ("flags", "u1"), ("combo", "<i2"), ("piece_count", "<u4"), ("seed", "<u4"),
    ("score", "<i4"), ("line_clears", "<i4")])
OBS_DTYPE = np.dtype([("rows", "<u2", (20,)), ("cells", "u1", (4,)), ("end", "u1"), ("pad", "u1", (3,))])
assert GAME_DTYPE.itemsize == 64 and OBS_DTYPE.itemsize == 48

EVAL_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p)
DIST_EVAL_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p)


def build(ref=True):
    """Compile the oracle (and, when /root/reference exists, the reference's own native sources)."""
    targets = ["own"] + (["ref"] if ref else [])
    subprocess.check_call(["make", "-s", "-C", HERE] + targets)


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(BUILD, "liboracle.so")
        if not os.path.exists(path):
            build(ref=os.path.isdir("/root/reference"))
        L = C.CDLL(path)
        vp, i32, f64, u32 = C.c_void_p, C.c_int, C.c_double, C.c_uint32
        L.orc_agent_new.restype = vp
        L.orc_agent_new.argtypes = [i32, i32, i32, i32, i32, f64, i32, i32, i32, i32, i32, vp, vp]
        L.orc_agent_free.argtypes = [vp]
        L.orc_agent_update_root.argtypes = [vp, vp]
        L.orc_agent_play.argtypes = [vp, i32]
        L.orc_agent_play.restype = i32
        L.orc_agent_compute_stats.argtypes = [vp, i32, vp]
        L.orc_agent_new_node.restype, L.orc_agent_new_node.argtypes = i32, [vp, vp]
        L.orc_agent_expand_game.argtypes = [vp, vp]
        L.orc_agent_remove_nodes.argtypes = [vp]
        for name in ("child", "score", "n_to_o", "visit", "value", "variance", "end_obs", "obs_state", "games",
                     "stats", "mem_state", "mem_value", "mem_variance", "mem_visit", "mem_dist", "rng"):
            f = getattr(L, "orc_agent_" + name)
            f.restype, f.argtypes = vp, [vp]
        for name in ("root", "episode", "error", "n_avail", "n_obs_avail", "memory_index"):
            f = getattr(L, "orc_agent_" + name)
            f.restype, f.argtypes = i32, [vp]
        for name in ("n_sims", "n_expand", "n_gc", "n_eval_states", "trace_len_sum", "max_trace_len", "n_eval_used", "n_eval_repeat",
                     "sweep_mismatch", "sweep_passes", "sweep_readings", "sweep_rows", "sweep_max_passes"):
            f = getattr(L, "orc_agent_" + name)
            f.restype, f.argtypes = C.c_long, [vp]
        L.orc_agent_set_mt.argtypes = [vp, vp]
        L.orc_agent_set_cpp_occupied.argtypes = [vp, i32]
        L.orc_agent_n_rollout_steps.restype, L.orc_agent_n_rollout_steps.argtypes = C.c_long, [vp]
        L.orc_mt_randint7_test.argtypes = [vp]
        L.orc_game_init.argtypes = [vp, i32, i32, i32, u32]
        L.orc_game_play.argtypes = [vp, i32, i32, i32, i32, vp]
        L.orc_game_reset.argtypes = [vp, i32, i32, i32, vp]
        L.orc_game_render.argtypes = [vp, vp]
        L.orc_game_pack_obs.argtypes = [vp, vp]
        L.orc_obs_render.argtypes = [vp, vp]
        L.orc_game_hash.restype = C.c_uint64
        L.orc_game_hash.argtypes = [vp]
        L.orc_obs_hash.restype = C.c_uint64
        L.orc_obs_hash.argtypes = [vp]
        L.orc_piece_at.argtypes = [u32, u32, i32]
        L.orc_srand.argtypes = [vp, u32]
        L.orc_rand.argtypes = [vp]
        L.orc_rand.restype = i32
        L.orc_norm_quantile.restype = f64
        L.orc_norm_quantile.argtypes = [f64]
        L.orc_exp.restype = f64
        L.orc_exp.argtypes = [f64]
        L.orc_get_unique_child_obs.argtypes = [i32, vp, vp, vp, vp, vp]
        L.orc_select_trace_obs.argtypes = [i32, vp, vp, vp, vp, vp, vp, i32, vp, vp, i32]
        L.orc_backup_trace_obs.argtypes = [vp, i32, vp, vp, vp, vp, vp, f64, f64, f64]
        L.orc_backup_trace_mixture_obs.argtypes = [vp, i32, vp, vp, vp, vp, vp, f64, f64, f64]
        L.orc_backup_trace_obs_LP.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, f64, i32, i32]
        L.orc_backup_obs_cppagent.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, f64, C.c_float]
        L.orc_get_all_childs.argtypes = [i32, vp, i32, vp]
        L.orc_sweep_marks.argtypes = [i32, vp, i32, vp, i32, vp]
        L.orc_valuenet_forward.argtypes = [vp, vp, i32, vp, vp]
        L.orc_hash_eval.argtypes = [vp, vp, i32, vp, vp]
        L.orc_hash_dist.argtypes = [vp, vp, i32, i32, vp]
        L.orc_distnet_forward.argtypes = [vp, vp, i32, i32, vp]
        L.orc_agent_set_dist.argtypes = [vp, i32, f64, f64, vp]
        L.orc_agent_node_stats.restype = L.orc_agent_node_dist.restype = vp
        L.orc_agent_node_stats.argtypes = L.orc_agent_node_dist.argtypes = [vp]
        L.orc_transform_distribution.argtypes = [vp, i32, f64, f64, f64, f64, vp]
        L.orc_mean_dist.restype, L.orc_mean_dist.argtypes = f64, [vp, i32, f64, f64]
        L.orc_mean_variance_dist.argtypes = [vp, i32, f64, f64, vp]
        L.orc_distpy_shift.argtypes = [vp, i32, f64, f64, f64, vp]
        L.orc_distpy_policy.restype, L.orc_distpy_policy.argtypes = i32, [vp, i32, vp, f64]
        L.orc_distpy_backup.argtypes = [vp, i32, vp, vp, i32, f64, vp, f64, f64, vp]
        _lib = L
    return _lib


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Game:
    """One oracle environment (ENGINE_SPEC.md) driven through liboracle."""

    def __init__(self, app=1, scoring=0, randomizer=0, seed=0):
        self.cfg = (app, scoring, randomizer)
        self.g = np.zeros(1, GAME_DTYPE)
        self.line_stats = np.zeros(4, np.int32)
        lib().orc_game_init(ptr(self.g), app, scoring, randomizer, seed)

    def play(self, a):
        lib().orc_game_play(ptr(self.g), *self.cfg, int(a), ptr(self.line_stats))

    def reset(self):
        lib().orc_game_reset(ptr(self.g), *self.cfg, ptr(self.line_stats))

    def copy_from(self, other):
        self.cfg = other.cfg
        self.g[...] = other.g
        self.line_stats[...] = other.line_stats

    def copy(self):
        o = Game.__new__(Game)
        o.cfg, o.g, o.line_stats = self.cfg, self.g.copy(), self.line_stats.copy()
        return o

    def getState(self):
        out = np.zeros((20, 10), np.int8)
        lib().orc_game_render(ptr(self.g), ptr(out))
        return out

    def packed_obs(self):
        o = np.zeros(1, OBS_DTYPE)
        lib().orc_game_pack_obs(ptr(self.g), ptr(o))
        return o

    end = property(lambda s: bool(s.g["flags"][0] & 1))
    score = property(lambda s: int(s.g["score"][0]))
    line_clears = property(lambda s: int(s.g["line_clears"][0]))
    combo = property(lambda s: int(s.g["combo"][0]))


class Agent:
    """Oracle tree agent. kind: 0 ValueSim, 1 ValueSimLP, 2 all-C++ agent LP, 3 all-C++ agent single, 4 Vanilla,
    5 VanillaC (the all-C++ agent with evaluator type 1 = random playout)."""

    def __init__(self, kind, max_nodes=100000, app=1, scoring=0, randomizer=0, gamma=0.999, low=1, benchmark=False,
                 online=False, min_visits_to_store=None, memory_size=0, evaluator="hash", params=None,
                 cpp_occupied=False, dist_bins=50, dist_vmin=0.0, dist_vmax=5000.0):
        L = lib()
        self.L = L
        if min_visits_to_store is None:
            min_visits_to_store = {0: 10, 1: 25}.get(kind, 40)
        self._keep = None
        if kind == 6 and evaluator == "distnet":
            # the distributional head (distnet_oracle.c); params: TM_DISTNET_PARAMS(dist_bins) floats, state_dict order
            self._keep = np.ascontiguousarray(params, np.float32)
            assert self._keep.size == 279232 + 129 * int(dist_bins)
            fn, ctx = C.cast(L.orc_hash_eval, C.c_void_p), ptr(self._keep)
        elif evaluator == "hash" or kind == 6:
            fn, ctx = C.cast(L.orc_hash_eval, C.c_void_p), None
        elif evaluator == "valuenet":
            self._keep = np.ascontiguousarray(params, np.float32)
            assert self._keep.size == 478342
            fn, ctx = C.cast(L.orc_valuenet_eval, C.c_void_p), ptr(self._keep)
        else:  # python callable(states int8 [k,20,10]) -> (v[k], var[k])
            def _cb(_ctx, states, k, v, var):
                s = np.ctypeslib.as_array(C.cast(states, C.POINTER(C.c_int8)), (k, 20, 10))
                vv, vr = evaluator(s.copy())
                np.ctypeslib.as_array(C.cast(v, C.POINTER(C.c_float)), (k,))[:] = np.asarray(vv, np.float32).ravel()
                np.ctypeslib.as_array(C.cast(var, C.POINTER(C.c_float)), (k,))[:] = np.asarray(vr, np.float32).ravel()
            self._keep = EVAL_FN(_cb)
            fn, ctx = C.cast(self._keep, C.c_void_p), None
        self.max_nodes = max_nodes
        self.h = L.orc_agent_new(max_nodes, app, scoring, randomizer, kind, gamma, low, int(benchmark), int(online),
                                 min_visits_to_store, memory_size, fn, ctx)
        if kind == 6:
            # DistValueSim (agent_oracle.c kind 6): per-node statistics and distributions; evaluator "hash" = orc_hash_dist, or a
            # python callable(states int8 [k,20,10]) -> float32 [k, bins]
            self.bins, self.vrange = int(dist_bins), (float(dist_vmin), float(dist_vmax))
            if evaluator == "hash":
                dfn = C.cast(L.orc_hash_dist, C.c_void_p)
            elif evaluator == "distnet":
                dfn = C.cast(L.orc_distnet_eval, C.c_void_p)
            else:
                def _dcb(_ctx, states, k, bins, out):
                    st = np.ctypeslib.as_array(C.cast(states, C.POINTER(C.c_int8)), (k, 20, 10))
                    d = np.asarray(evaluator(st.copy()), np.float32).reshape(k, bins)
                    np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_float)), (k, bins))[:] = d
                self._keep_dist = DIST_EVAL_FN(_dcb)
                dfn = C.cast(self._keep_dist, C.c_void_p)
            L.orc_agent_set_dist(self.h, self.bins, self.vrange[0], self.vrange[1], dfn)
        if cpp_occupied:
            # the reference C++ agent's `occupied` vector slip (agent.cpp:300-301, see agent_oracle.c); only where the
            # oracle is compared with the reference's compiled agent, never where it checks the product
            assert kind in (2, 3, 5)
            L.orc_agent_set_cpp_occupied(self.h, 1)

    def set_python_random_state(self, state):
        """state = random.getstate() (or random.Random(seed).getstate()): the rollout RNG of Vanilla (Vanilla.py:4,52)."""
        st = np.asarray(state[1], dtype=np.uint64).astype(np.uint32)
        assert st.size == 625
        self.L.orc_agent_set_mt(self.h, ptr(st))

    def update_root(self, game):
        self.L.orc_agent_update_root(self.h, ptr(game.g))

    def play(self, sims):
        return self.L.orc_agent_play(self.h, sims)

    # TreeAgent's single calls (agent.cpp:212-218,265-270,337; agents/agent.py:90-145,246-257)
    def new_node(self, game):
        return self.L.orc_agent_new_node(self.h, ptr(game.g))

    def expand(self, game):
        self.L.orc_agent_expand_game(self.h, ptr(game.g))

    def remove_nodes(self):
        self.L.orc_agent_remove_nodes(self.h)

    def compute_stats(self):
        out = np.zeros((3, 7), np.float32)
        self.L.orc_agent_compute_stats(self.h, self.root, ptr(out))
        return out

    def _arr(self, name, dtype, shape):
        p = getattr(self.L, "orc_agent_" + name)(self.h)
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        buf = (C.c_char * n).from_address(p)
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    def arrays(self):
        n = self.max_nodes
        return dict(child=self._arr("child", np.int32, (n, 7)), score=self._arr("score", np.float32, (n,)),
                    n_to_o=self._arr("n_to_o", np.int32, (n,)), visit=self._arr("visit", np.int32, (n,)),
                    value=self._arr("value", np.float32, (n,)), variance=self._arr("variance", np.float32, (n,)),
                    end_obs=self._arr("end_obs", np.uint8, (n,)), obs_state=self._arr("obs_state", OBS_DTYPE, (n,)),
                    games=self._arr("games", GAME_DTYPE, (n,)))

    def stats(self):
        return self._arr("stats", np.float32, (3, 7)).copy()

    def dist_arrays(self):
        """kind 6: (node_stats [n,5] = visit, mean, score, variance, M2; node_dist [n,bins]) as core_distributional.py holds them"""
        n = self.max_nodes
        return self._arr("node_stats", np.float32, (n, 5)), self._arr("node_dist", np.float32, (n, self.bins))

    def memory(self):
        m = self.L.orc_agent_memory_index(self.h)
        if m == 0:
            return (np.zeros((0, 200), np.int8), np.zeros(0, np.float32), np.zeros(0, np.float32),
                    np.zeros(0, np.float32))
        return (self._arr("mem_state", np.int8, (m, 200)).copy(), self._arr("mem_value", np.float32, (m,)).copy(),
                self._arr("mem_variance", np.float32, (m,)).copy(), self._arr("mem_visit", np.float32, (m,)).copy())

    def memory_dist(self):
        """kind 6, online: (states int8 [m,200], distributions [m,bins], visits [m]) stored by the collections so far"""
        m = self.L.orc_agent_memory_index(self.h)
        if m == 0:
            return np.zeros((0, 200), np.int8), np.zeros((0, self.bins), np.float32), np.zeros(0, np.float32)
        return (self._arr("mem_state", np.int8, (m, 200)).copy(), self._arr("mem_dist", np.float32, (m, self.bins)).copy(),
                self._arr("mem_visit", np.float32, (m,)).copy())

    def __getattr__(self, name):
        if name in ("root", "episode", "error", "n_avail", "n_obs_avail", "memory_index", "n_sims", "n_expand",
                    "n_gc", "n_eval_states", "trace_len_sum", "max_trace_len", "n_rollout_steps", "n_eval_used", "n_eval_repeat",
                    "sweep_mismatch", "sweep_passes", "sweep_readings", "sweep_rows", "sweep_max_passes"):
            return getattr(self.L, "orc_agent_" + name)(self.h)
        raise AttributeError(name)

    def close(self):
        if self.h:
            self.L.orc_agent_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def oracle_pytetris():
    """The oracle's compiled `pyTetris` extension (oracle/_build/pyTetris*.so), loaded by path: the product ships a
    module of the same name (the reference's import line must resolve to the GPU engine), so `import pyTetris` may
    return either one depending on sys.path and on what the process imported first.  Checker code never relies on
    that: it asks here.  sys.modules is left alone."""
    import importlib.util
    import sysconfig
    path = os.path.join(BUILD, "pyTetris" + sysconfig.get_config_var("EXT_SUFFIX"))
    m = sys.modules.get("pyTetris")
    if m is not None and os.path.abspath(getattr(m, "__file__", "") or "") == os.path.abspath(path):
        return m
    cached = getattr(oracle_pytetris, "_mod", None)
    if cached is None:
        if not os.path.exists(path):
            raise RuntimeError("%s is missing: make -C oracle own" % path)
        spec = importlib.util.spec_from_file_location("pyTetris", path)
        cached = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(cached)
        oracle_pytetris._mod = cached
    return cached


def load_ref_native(name):
    """Import the reference's own compiled module (oracle/_ref/<name>*.so): 'core' or 'agent'."""
    import importlib.util
    import sysconfig
    oracle_pytetris()          # the agent module needs the oracle's Tetris type registered (pybind, per interpreter)
    path = os.path.join(REFOUT, name + sysconfig.get_config_var("EXT_SUFFIX"))
    if not os.path.exists(path):
        return None
    full = "agents.cppmodule." + name
    if full in sys.modules:
        return sys.modules[full]
    spec = importlib.util.spec_from_file_location(full, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod
