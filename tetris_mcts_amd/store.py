"""Device-resident tree store (include/tetris_mcts_hip.h `tm_store`) and thin launch wrappers.

PyTorch is plumbing here: device memory, the current HIP stream, and torch.distributed for sharding.
All compute on the path goes through libtetris_mcts_hip.so.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

SIM_BACKUP, SIM_FRONT, SIM_GC_FULL, SIM_EVAL_NEEDED = 1, 2, 4, 8
KIND_VALUESIM, KIND_VALUESIM_LP, KIND_CPPAGENT_LP, KIND_CPPAGENT, KIND_VANILLA, KIND_VANILLA_C, KIND_DIST = 0, 1, 2, 3, 4, 5, 6
GS = dict(ROOT=0, EPISODE=1, NFREE_NODE=2, NFREE_OBS=3, TRACE_LEN=4, PENDING=5, ERR=6, N_EXPAND=7, N_SIMS=8, N_GC=9,
          RNG_POS=10, N_NQ_FALLBACK=11, LEAF=12, LEAF_END=13, K_EVAL=14, LEAF_SCORE=15, TRACE_SUM=16, N_EVAL=17, N_POOL_RESET=18, MAX_TRACE=19, N_DROPPED=25,
          CYC_BACK=20, CYC_SELECT=21, CYC_EXPAND=22, CYC_GC16=23, GC_REACHABLE=24, GC_PHASE=32, GC_ACTIVE4=34, GC_WORK=36, GC_MARK_LAUNCHES=37, GC_SLICES=38,
          SIM_TARGET=40, SIM_STARTED=41, CYC_VERIFY=42, FIRST_MISS=28, PREFIX_SUM=29, N_WALK_MISS=43, POOL_FULL=44, GC_IN_MOVE=45, GC_REQ_AT=46,
          LEAF_OBS=47, N_EVAL_SKIP=48, N_EVAL_CACHED=49, GC_NGC=50, GC_BLOCKS=51, GC_ITERS=52, GC_MARK_CYC=53, GC_FLAGS=54, GC_MARK_PARTS=58, GC_MARK_SHARED=59, GC_CYC_LOAD=60, GC_CYC_ROUNDS=61, GC_CYC_WAVES=62, GC_IDLE_TURNS=63)

_nq_cache = {}


def norm_quantile_table(n, device):
    """(float)norm_quantile(i), special.h:26-33, evaluated with the host libm like the reference does."""
    key = (n, str(device))
    if key not in _nq_cache:
        host = np.zeros(n, np.float32)
        with np.errstate(all="ignore"):
            _lib.lib().tm_fill_norm_quantile(host.ctypes.data_as(C.c_void_p), n)
        _nq_cache[key] = torch.from_numpy(host).to(device)
    return _nq_cache[key]


def norm_quantile_table_f64(n, device):
    """norm_quantile(i) in double (policy_dist of the distributional agent multiplies in double)"""
    key = ("f64", n, str(device))
    if key not in _nq_cache:
        host = np.zeros(n, np.float64)
        with np.errstate(all="ignore"):
            _lib.lib().tm_fill_norm_quantile_f64(host.ctypes.data_as(C.c_void_p), n)
        _nq_cache[key] = torch.from_numpy(host).to(device)
    return _nq_cache[key]


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class TreeStore:
    """Node / observation pools, transposition tables and per-game control blocks for `n_games` games."""

    def __init__(self, n_games, max_nodes=100000, kind=KIND_VALUESIM, env_args=((20, 10), 1, 0, 0), gamma=0.999,
                 low=1, eval_slots=None, max_trace=1024, nq_size=1 << 20, online=False, min_visits_to_store=10,
                 replay_cap=0, gc_slice_cycles=150000, gc_spec_nodes=None, gc_cost_units=0, gc_collectors=0, dist_bins=50, dist_range=(0.0, 5000.0), device="cuda"):
        if not torch.cuda.is_available():
            raise RuntimeError("tetris_mcts_amd needs a ROCm GPU (gfx950); there is no CPU path")
        shape, app, scoring, randomizer = env_args[0], env_args[1], env_args[2], env_args[3]
        if tuple(shape) != (20, 10):
            raise ValueError("only 20x10 boards are supported")
        if max_trace % 64:
            raise ValueError("max_trace must be a multiple of 64 (the walk flushes its LDS trace buffer every 64 levels)")
        self.L = _lib.lib()
        self.device = torch.device(device)
        self.n_games, self.max_nodes, self.kind = int(n_games), int(max_nodes), int(kind)
        if eval_slots is None:
            eval_slots = 7 if kind in (KIND_VALUESIM_LP, KIND_CPPAGENT_LP) else 1
        self.eval_slots = eval_slots
        cap = 16
        while cap < 2 * max_nodes:
            cap <<= 1
        G, N, dev = self.n_games, self.max_nodes, self.device
        z = lambda *shape, dtype=torch.int32: torch.zeros(*shape, dtype=dtype, device=dev)  # noqa: E731
        bm = ((N + 7) // 8 + 15) & ~15
        self.t = dict(
            node_rec=z(G, N, 32), node_child=z(G, N, 8), node_game=z(G, N, 16), obs_stat=z(G, N, 4), obs_key=z(G, N, 12),
            node_tab=z(G, cap, dtype=torch.int64), obs_tab=z(G, cap, dtype=torch.int64),
            free_node=z(G, N), free_obs=z(G, N), gs=z(G, 64), rng=z(G, 32), env_game=z(G, 16), env_line_stats=z(G, 4),
            trace=z(G, max_trace, 4), leaf=z(G, 32), eval_obs=z(G * eval_slots),
            eval_v=z(G * eval_slots, dtype=torch.float32), eval_var=z(G * eval_slots, dtype=torch.float32),
            gc_mark=z(G, 2 * bm, dtype=torch.uint8), gc_queue=z(G, N), gc_part=z(G, 192),
            replay_obs=z(G, max(replay_cap, 1), 12), replay_stat=z(G, max(replay_cap, 1), 4, dtype=torch.float32),
            replay_count=z(G),
            mt_state=z(G if kind in (KIND_VANILLA, KIND_VANILLA_C) else 1, 625),
        )
        # the dense request list of every launch, and (single-leaf kinds with an observation projection) the evaluator's output
        # per observation - include/tetris_mcts_hip.h tm_store::eval_list / obs_eval
        self.t["eval_list"] = z(G * eval_slots, 2)
        self.t["eval_cnt"] = z(G, 2)
        self.t["obs_eval"] = z(G, N, 4, dtype=torch.float32) if kind in (KIND_VALUESIM, KIND_CPPAGENT) else None
        # the distributional agent's harvest (DistValueSimOnline.store_nodes): the freed nodes' distributions
        self.t["replay_dist"] = z(G, replay_cap, 64, dtype=torch.float32) if (kind == KIND_DIST and online and replay_cap > 0) else None
        self.t["game_list"] = None       # (tm_store::game_list: the native loop's catch-up launches set it in their own copies)
        self.t["nq_table"] = norm_quantile_table(nq_size, dev)
        # TM_KIND_DIST only: the nodes' value distributions, the evaluator's output, norm_quantile in double
        is_dist = kind == KIND_DIST
        if is_dist and not (0 < dist_bins <= 64):
            raise ValueError("1..64 atoms")
        self.t["node_dist"] = z(G, N if is_dist else 1, 64, dtype=torch.float32)
        self.t["eval_dist"] = z(G, 64, dtype=torch.float32)
        self.t["nq_table_d"] = norm_quantile_table_f64(nq_size, dev) if is_dist else torch.zeros(1, dtype=torch.float64, device=dev)
        s = _lib.TmStore()
        s.n_games, s.max_nodes, s.table_cap, s.max_trace, s.eval_slots, s.nq_size = G, N, cap, max_trace, eval_slots, nq_size
        s.app, s.scoring, s.randomizer = int(app), int(scoring), int(randomizer)
        s.low, s.kind, s.min_visits_to_store, s.online, s.replay_cap = int(low), int(kind), int(min_visits_to_store), int(bool(online)), int(replay_cap)
        s.gc_slice_cycles = int(gc_slice_cycles)
        # speculative marking (tree.hip GC_SPEC_*) starts when a game has fewer free nodes than this
        s.gc_spec_nodes = int(min(256, self.max_nodes // 8) if gc_spec_nodes is None else gc_spec_nodes)
        s.gamma = float(gamma)
        for name, typ in _lib.TmStore._fields_[16:]:
            if typ is C.c_void_p:
                setattr(s, name, self.t[name].data_ptr() if self.t[name] is not None else None)
        s.dist_bins, s.dist_vmin, s.dist_vmax = int(dist_bins), float(dist_range[0]), float(dist_range[1])
        s.eval_parity, s.eval_epoch, s.n_listed, s.gc_cost_units, s.gc_collectors = 0, 0, 0, int(gc_cost_units or 0), int(gc_collectors or 0)
        self.s = s
        self.stats_buf = torch.zeros(G, 3, 7, dtype=torch.float32, device=dev)
        self.action_buf = torch.zeros(G, dtype=torch.int32, device=dev)
        self.eval_states = torch.zeros(G * eval_slots, 200, dtype=torch.int8, device=dev)
        self._rem = torch.zeros(2, dtype=torch.int32, device=dev)
        self._search = {}
        _lib.check(self.L.tm_pool_init(C.byref(s), _stream()), "tm_pool_init")

    def __del__(self):
        for h in getattr(self, "_search", {}).values():
            self.L.tm_search_destroy(h)
        self._search = {}

    def set_python_random_states(self, states):
        """states: one `random.getstate()` / `random.Random(seed).getstate()` per game (Vanilla's rollout RNG)."""
        arr = np.stack([np.asarray(st[1], dtype=np.uint64).astype(np.uint32) for st in states])
        assert arr.shape == (self.n_games, 625)
        self.t["mt_state"].copy_(torch.from_numpy(arr.view(np.int32)))

    def nbytes(self):
        return sum(t.numel() * t.element_size() for t in self.t.values() if t is not None)

    # ---- tree agent entry points ----
    def set_root_games(self, games):
        """games: uint32/int32 [G,16] packed games (device)."""
        self.t["env_game"].copy_(games.view(torch.int32))

    def update_root(self):
        _lib.check(self.L.tm_update_root(C.byref(self.s), _stream()), "tm_update_root")

    # TreeAgent's single calls (agent.cpp:828-833; agents/agent.py:90-145,246-257), one game state per tree
    def _mask(self, mask):
        if mask is None:
            return None, C.c_void_p(0)
        m = torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
        return m, _p(m)

    def new_node(self, games, mask=None, expand=False):
        """games: [G,16] packed games (device).  Returns the node index of every game's state (int32 [G], device);
        expand=True also links its seven successors (TreeAgent.expand)."""
        g = games.view(torch.int32).contiguous()
        assert g.shape == (self.n_games, 16) and g.device.type == "cuda"
        out = torch.empty(self.n_games, dtype=torch.int32, device=self.device)
        keep, mp = self._mask(mask)
        fn = self.L.tm_tree_expand if expand else self.L.tm_tree_new_node
        _lib.check(fn(C.byref(self.s), _p(g), mp, _p(out), _stream()), "tm_tree_expand" if expand else "tm_tree_new_node")
        return out

    def remove_nodes(self, mask=None):
        keep, mp = self._mask(mask)
        _lib.check(self.L.tm_tree_remove_nodes(C.byref(self.s), mp, _stream()), "tm_tree_remove_nodes")

    def move_begin(self, sims):
        """Give every game a quota of `sims` more simulations (TreeAgent.play: mcts(root, sims), agents/agent.py:147-150)."""
        _lib.check(self.L.tm_move_begin(C.byref(self.s), int(sims), _stream()), "tm_move_begin")

    def sims_remaining(self):
        """(launches still needed before every game has used its quota, games whose collection is under way) - a host
        sync; the first is > 0 only when a game spent launches collecting garbage instead of simulating."""
        _lib.check(self.L.tm_sims_remaining(C.byref(self.s), _p(self._rem), _stream()), "tm_sims_remaining")
        r = self._rem.cpu().numpy()
        return int(r[0]), int(r[1])

    def gc_step(self):
        """One step of every garbage collection under way (collector workgroups only)."""
        _lib.check(self.L.tm_gc_step(C.byref(self.s), _stream()), "tm_gc_step")

    def search(self, sims, model=None, n_sub=1, ev_every=0):
        """One move's search through the native launch loop (search.hip): `sims` simulations of every game, the leaf
        evaluator = the HIP value net of `model` (None: no evaluator, the Vanilla kind).  Returns when complete."""
        key = (int(n_sub), int(ev_every))
        h = self._search.get(key)
        if h is None:
            h = C.c_void_p()
            _lib.check(self.L.tm_search_create(C.byref(h), C.byref(self.s), key[0], key[1]), "tm_search_create")
            self._search[key] = h
        if model is None:
            P = prep = scr = C.c_void_p(0)
        else:
            P, prep, scr = model.hip_buffers(self.n_games * self.eval_slots)
            # the version of the weights the evaluator runs on: what was filed per observation under another one is not used
            self.L.tm_search_set_epoch(h, int(getattr(model, "weights_epoch", 0)))
        _lib.check(self.L.tm_search_run(h, int(sims), P, prep, scr, _stream()), "tm_search_run")

    def search_stats(self, n_sub=1, ev_every=0, reset=True):
        """dict of the native loop's counters and HIP-event timings (tm_search_stats)."""
        h = self._search.get((int(n_sub), int(ev_every)))
        if h is None:
            return None
        out = (C.c_double * 11)()
        self.L.tm_search_stats(h, out, 11, int(bool(reset)))
        keys = ("runs", "tree_launches", "catchup_launches", "timed", "tree_ms_sum", "nn_ms_sum", "n_sub", "gc_launches",
                "loop_ms_sum", "loop_sims", "catchup_waves")
        return dict(zip(keys, list(out)))

    def sim_step(self, flags):
        if flags & SIM_FRONT:
            self.s.eval_parity ^= 1      # the dense request list: a launch appends under the other parity (tm_store::eval_list)
        _lib.check(self.L.tm_sim_step(C.byref(self.s), int(flags), _stream()), "tm_sim_step")

    def render_eval(self):
        _lib.check(self.L.tm_eval_render(C.byref(self.s), _p(self.eval_states), _stream()), "tm_eval_render")
        return self.eval_states

    def root_stats(self):
        _lib.check(self.L.tm_root_stats(C.byref(self.s), _p(self.stats_buf), _p(self.action_buf), _stream()),
                   "tm_root_stats")
        return self.stats_buf, self.action_buf

    def pool_reset(self, mask):
        """Re-initialise the trees of the games where mask is true (uint8/bool [G], device)."""
        m = mask.to(torch.uint8).contiguous()
        _lib.check(self.L.tm_pool_reset(C.byref(self.s), _p(m), _stream()), "tm_pool_reset")

    def errors(self):
        return self.t["gs"][:, GS["ERR"]]

    def counter(self, name):
        return int(self.t["gs"][:, GS[name]].sum().item())

    def export_game(self, g):
        """One game's tree in the reference's array layout (agents/agent.py:58-88), as numpy arrays."""
        N, dev = self.max_nodes, self.device
        out = dict(child=torch.zeros(N, 7, dtype=torch.int32, device=dev), score=torch.zeros(N, device=dev),
                   n_to_o=torch.zeros(N, dtype=torch.int32, device=dev), visit=torch.zeros(N, dtype=torch.int32, device=dev),
                   value=torch.zeros(N, device=dev), variance=torch.zeros(N, device=dev),
                   end_obs=torch.zeros(N, dtype=torch.uint8, device=dev))
        _lib.check(self.L.tm_export_game(C.byref(self.s), int(g), _p(out["child"]), _p(out["score"]), _p(out["n_to_o"]),
                                         _p(out["visit"]), _p(out["value"]), _p(out["variance"]), _p(out["end_obs"]),
                                         _stream()), "tm_export_game")
        return {k: v.cpu().numpy() for k, v in out.items()}

    def replay(self):
        """Replay tuples harvested by GC (ValueSim.store_nodes, ValueSim.py:122-159): packed obs, value, variance, visit."""
        cnt = self.t["replay_count"].clamp(max=self.s.replay_cap)
        idx = torch.arange(self.t["replay_obs"].shape[1], device=self.device)[None, :] < cnt[:, None]
        return self.t["replay_obs"][idx], self.t["replay_stat"][idx]

    def replay_dist(self):
        """TM_KIND_DIST: (packed observations, distributions [n, 64], visit counts) of the nodes harvested by GC
        (DistValueSimOnline.store_nodes, agents/DistValueSimOnline.py:116-141), per game in index order."""
        cnt = self.t["replay_count"].clamp(max=self.s.replay_cap)
        idx = torch.arange(self.t["replay_obs"].shape[1], device=self.device)[None, :] < cnt[:, None]
        return self.t["replay_obs"][idx], self.t["replay_dist"][idx], self.t["replay_stat"][idx][:, 2]
