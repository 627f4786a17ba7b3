"""ctypes binding of libtetris_mcts_hip.so (include/tetris_mcts_hip.h).

The HIP library IS the product: there is no CPU fallback.  If the shared object is missing the import
fails loudly and tells the user to run `python -c "import __graft_entry__ as g; g.build()"`.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TETRIS_MCTS_LIB", os.path.join(HERE, "libtetris_mcts_hip.so"))

vp, i32, f64 = C.c_void_p, C.c_int32, C.c_double


class TmStore(C.Structure):
    """Mirror of `tm_store` (include/tetris_mcts_hip.h); layout checked by tests/test_abi.py."""
    _fields_ = [
        ("n_games", i32), ("max_nodes", i32), ("table_cap", i32), ("max_trace", i32), ("eval_slots", i32),
        ("nq_size", i32), ("app", i32), ("scoring", i32), ("randomizer", i32), ("low", i32), ("kind", i32),
        ("min_visits_to_store", i32), ("online", i32), ("replay_cap", i32), ("gc_slice_cycles", i32), ("gamma", f64),
        ("node_rec", vp), ("node_game", vp), ("obs_stat", vp), ("obs_key", vp), ("node_tab", vp), ("obs_tab", vp),
        ("free_node", vp), ("free_obs", vp), ("gs", vp), ("rng", vp), ("env_game", vp), ("env_line_stats", vp),
        ("trace", vp), ("leaf", vp), ("eval_obs", vp), ("eval_v", vp), ("eval_var", vp), ("nq_table", vp),
        ("gc_mark", vp), ("gc_queue", vp), ("replay_obs", vp), ("replay_stat", vp), ("replay_count", vp), ("mt_state", vp), ("node_child", vp), ("gc_part", vp),
        ("node_dist", vp), ("eval_dist", vp), ("nq_table_d", vp), ("dist_vmin", f64), ("dist_vmax", f64), ("dist_bins", i32),
        ("gc_spec_nodes", i32),
        ("eval_list", vp), ("eval_cnt", vp), ("obs_eval", vp), ("eval_parity", i32), ("eval_epoch", i32),
        ("replay_dist", vp), ("game_list", vp), ("n_listed", i32), ("gc_cost_units", i32), ("gc_collectors", i32),
    ]


# every symbol include/tetris_mcts_hip.h declares
SYMBOLS = {
    "tm_pool_init": [C.POINTER(TmStore), vp],
    "tm_pool_reset": [C.POINTER(TmStore), vp, vp],
    "tm_env_init": [C.POINTER(TmStore), vp, vp],
    "tm_env_step": [C.POINTER(TmStore), vp, vp],
    "tm_env_reset": [C.POINTER(TmStore), vp, vp],
    "tm_env_render": [C.POINTER(TmStore), vp, vp],
    "tm_env_info": [C.POINTER(TmStore), vp, vp],
    "tm_update_root": [C.POINTER(TmStore), vp],
    "tm_tree_new_node": [C.POINTER(TmStore), vp, vp, vp, vp],
    "tm_tree_expand": [C.POINTER(TmStore), vp, vp, vp, vp],
    "tm_tree_remove_nodes": [C.POINTER(TmStore), vp, vp],
    "tm_sim_step": [C.POINTER(TmStore), i32, vp],
    "tm_move_begin": [C.POINTER(TmStore), i32, vp],
    "tm_sims_remaining": [C.POINTER(TmStore), vp, vp],
    "tm_sims_owing": [C.POINTER(TmStore), vp, vp, vp],
    "tm_gc_step": [C.POINTER(TmStore), vp],
    "tm_eval_render": [C.POINTER(TmStore), vp, vp],
    "tm_root_stats": [C.POINTER(TmStore), vp, vp, vp],
    "tm_export_game": [C.POINTER(TmStore), i32, vp, vp, vp, vp, vp, vp, vp, vp],
    "tm_core_select_trace_obs": [i32, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, i32, vp, vp, i32, vp],
    "tm_core_backup_trace_obs": [i32, i32, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, f64, vp],
    "tm_core_backup_trace_obs_lp": [i32, i32, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, f64, i32, i32, vp],
    "tm_core_get_unique_child_obs": [i32, i32, vp, vp, vp, vp, vp, vp, vp, vp],
    "tm_core_get_all_childs": [i32, i32, vp, vp, vp, vp, vp],
    "tm_yogi_step": [vp, vp, vp, vp, vp, i32, f64, f64, f64, f64, f64, vp],
    "tm_valuenet_prepare": [vp, vp, vp],
    "tm_valuenet_forward": [vp, vp, vp, i32, vp, vp, vp, vp],
    "tm_valuenet_forward_plain": [vp, vp, i32, vp, vp, vp, vp],
    "tm_valuenet_forward_requests": [vp, vp, C.POINTER(TmStore), vp, vp],
    "tm_distnet_prepare": [vp, vp, vp],
    "tm_distnet_forward": [vp, vp, vp, i32, i32, vp, i32, vp, vp],
    "tm_distnet_forward_requests": [vp, vp, C.POINTER(TmStore), vp, vp],
    "tm_dist_transform": [i32, i32, vp, f64, f64, vp, f64, vp, vp],
    "tm_dist_mean_variance": [i32, i32, vp, f64, f64, vp, vp],
    "tm_distpy_shift": [i32, i32, vp, vp, f64, f64, vp, vp],
    "tm_distpy_policy": [i32, i32, vp, vp, vp, vp, vp, vp],
    "tm_distpy_backup": [i32, i32, i32, vp, vp, i32, vp, vp, vp, vp, f64, f64, vp, vp],
    "tm_store_slice": [C.POINTER(TmStore), i32, i32, C.POINTER(TmStore)],
    "tm_search_create": [C.POINTER(vp), C.POINTER(TmStore), i32, i32],
    "tm_search_run": [vp, i32, vp, vp, vp, vp],
    "tm_search_stats": [vp, vp, i32, i32],
    "tm_search_set_epoch": [vp, i32],
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "tetris_mcts_amd: %s is missing. This package has no CPU path; build the HIP library first:\n"
                "    python -c 'import __graft_entry__ as g; g.build()'" % LIB_PATH)
        # PyTorch ships its own libamdhip64; it must be the HIP runtime of this process (device memory, streams and
        # torch.distributed all go through it), so it is loaded first and our library binds to the same copy.
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, args in SYMBOLS.items():
            f = getattr(L, name)
            f.argtypes, f.restype = args, i32
        L.tm_fill_norm_quantile.argtypes, L.tm_fill_norm_quantile.restype = [vp, i32], None
        L.tm_fill_norm_quantile_f64.argtypes, L.tm_fill_norm_quantile_f64.restype = [vp, i32], None
        L.tm_version.argtypes, L.tm_version.restype = [], C.c_char_p
        L.tm_store_layout.argtypes, L.tm_store_layout.restype = [vp, i32], i32
        L.tm_search_destroy.argtypes, L.tm_search_destroy.restype = [vp], None
        _lib = L
    return _lib


def check(err, what):
    if err != 0:
        raise RuntimeError("%s failed with hipError %d" % (what, err))
