"""The C-ABI library loads without a GPU and exports every symbol include/tetris_mcts_hip.h declares."""
import ctypes as C
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    import __graft_entry__ as ge
    if not os.path.exists(ge.LIB):
        ge.build()
    from tetris_mcts_amd import _lib
    return _lib


def test_every_declared_symbol_is_exported():
    L = _lib()
    lib = L.lib()
    hdr = open(os.path.join(ROOT, "include", "tetris_mcts_hip.h")).read()
    declared = set(re.findall(r"\b(tm_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 19
    for name in declared:
        assert hasattr(lib, name), name
    assert set(L.SYMBOLS) <= declared
    assert lib.tm_version().startswith(b"tetris_mcts_hip")


def test_store_layout_matches_ctypes_mirror():
    L = _lib()
    out = (C.c_int * 10)()
    n = L.lib().tm_store_layout(out, 10)
    T = L.TmStore
    assert n == 10 and list(out)[:n] == [C.sizeof(T), T.gamma.offset, T.node_rec.offset, T.nq_table.offset,
                                         T.replay_count.offset, T.mt_state.offset, T.node_child.offset,
                                         T.gc_slice_cycles.offset, T.gc_part.offset, T.dist_vmin.offset]


def test_norm_quantile_table_is_the_reference_formula(oracle):
    L = _lib()
    t = np.zeros(4096, np.float32)
    with np.errstate(all="ignore"):
        L.lib().tm_fill_norm_quantile(t.ctypes.data_as(C.c_void_p), len(t))
    assert np.isnan(t[1]) and t[2] == 0.0
    for n in (3, 7, 100, 4095):
        assert t[n] == np.float32(oracle.lib().orc_norm_quantile(float(n)))


def test_missing_library_fails_loudly(monkeypatch):
    L = _lib()
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", "/nonexistent/libtetris_mcts_hip.so")
    try:
        L.lib()
    except RuntimeError as e:
        assert "no CPU path" in str(e)
    else:
        raise AssertionError("expected a loud failure")


def test_play_cli_keeps_the_reference_flags():
    """play.py:46-70 of the reference: every flag, its type and default survive."""
    import play
    p = play.build_parser()
    ref = dict(agent_type=None, app=1, benchmark=False, cycle=0, endless=False, gamma=0.9, gui=False, interactive=False,
               mcts_const=5.0, mcts_sims=50, mcts_tau=1.0, min_visit=40, ngames=50, online=False, printboard=False,
               print_board_to_file=False, realtime_status=False, save=False, save_dir='./data/', save_file='data',
               save_tree=False, tetris_randomizer=0, tetris_scoring=0)
    d = vars(p.parse_args([]))
    for k, v in ref.items():
        assert d[k] == v and type(d[k]) is type(v), k


def _is_message(call):
    """print / raise / warn texts may name the oracle."""
    import ast
    f = call.func
    name = f.id if isinstance(f, ast.Name) else getattr(f, "attr", "")
    return name in ("print", "warn", "RuntimeError", "ValueError", "NotImplementedError", "write", "format", "join")


def test_the_product_never_touches_the_oracle():
    """The oracle is the checker: nothing under tetris_mcts_amd/ (nor the root module shims, play.py, the C sources) may
    import, load or execute it; bench.py may only inside its cpu_baseline leg (functions named _cpu_* / cpu_baseline),
    __graft_entry__ only inside smoke()."""
    import ast
    product = []
    for base, _dirs, files in os.walk(os.path.join(ROOT, "tetris_mcts_amd")):
        product += [os.path.join(base, f) for f in files if f.endswith((".py", ".hip", ".h"))]
    product += [os.path.join(ROOT, f) for f in ("play.py", "pyTetris.py")]
    product += [os.path.join(ROOT, "agents", f) for f in os.listdir(os.path.join(ROOT, "agents")) if f.endswith(".py")]
    for path in product:
        text = open(path).read()
        if path.endswith(".py"):
            tree = ast.parse(text)
            for n in ast.walk(tree):
                if isinstance(n, ast.ImportFrom):
                    assert (n.module or "").split(".")[0] != "oracle", path
                if isinstance(n, ast.Import):
                    assert all(a.name.split(".")[0] != "oracle" for a in n.names), path
                if isinstance(n, ast.Call) and not _is_message(n):
                    # no path or module name with "oracle" in it handed to any call (CDLL, import_module, open ...)
                    for arg in list(n.args) + [k.value for k in n.keywords]:
                        names = [c.value for c in ast.walk(arg) if isinstance(c, ast.Constant) and isinstance(c.value, str)]
                        assert not any("oracle" in v for v in names), (path, n.lineno)
        else:
            assert not [ln for ln in text.splitlines() if "#include" in ln and "oracle" in ln], path

    def oracle_users(path):
        tree = ast.parse(open(path).read())
        users = set()
        for fn in [n for n in ast.walk(tree) if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef))]:
            for n in ast.walk(fn):
                if isinstance(n, ast.ImportFrom) and (n.module or "").split(".")[0] == "oracle":
                    users.add(fn.name)
                if isinstance(n, ast.Import) and any(a.name.split(".")[0] == "oracle" for a in n.names):
                    users.add(fn.name)
        top = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom))]
        assert not any((getattr(n, "module", None) or "").startswith("oracle") for n in top), path
        return users
    assert oracle_users(os.path.join(ROOT, "bench.py")) <= {"_cpu_worker", "_cpu_worker_vanilla", "cpu_baseline", "cpu_baseline_vanilla"}
    assert oracle_users(os.path.join(ROOT, "__graft_entry__.py")) <= {"smoke"}


def test_oracle_engine_is_loaded_by_path_whatever_pyTetris_means(oracle):
    """The product ships a module named `pyTetris` (the reference's import line must resolve to the GPU engine) and so
    does the oracle's build directory.  Checker code asks oracle.binding.oracle_pytetris() and gets the compiled oracle
    extension even when the product's module was imported first; sys.modules is not touched."""
    import importlib
    import sys
    sys.modules.pop("pyTetris", None)
    sys.path.insert(0, ROOT)
    try:
        prod = importlib.import_module("pyTetris")
        assert "oracle" not in (prod.__file__ or "")
        mod = oracle.oracle_pytetris()
        assert mod is not prod and mod.__file__.startswith(oracle.BUILD)
        assert sys.modules["pyTetris"] is prod
        g = mod.Tetris((20, 10), 1, 0, 0, 5)
        g.play(3)
        assert g.score >= 0 and oracle.oracle_pytetris() is mod
    finally:
        sys.path.remove(ROOT)
        sys.modules.pop("pyTetris", None)


def test_control_block_words_match_the_header():
    """store.GS (the Python names of the per-game control words) against the enum of include/tetris_mcts_hip.h."""
    import importlib
    hdr = open(os.path.join(ROOT, "include", "tetris_mcts_hip.h")).read()
    body = hdr[hdr.index("enum {", hdr.index("per-game control block (int32 words)")):]
    body = re.sub(r"/\*.*?\*/", "", body[:body.index("};")], flags=re.S)
    words, nxt = {}, 0
    for item in body[body.index("{") + 1:].split(","):
        item = item.strip()
        if not item:
            continue
        m = re.match(r"(TM_GS_[A-Z0-9_]+)(?:\s*=\s*(\d+))?$", item)
        assert m, item
        nxt = int(m.group(2)) if m.group(2) else nxt
        words[m.group(1)[6:]] = nxt
        nxt += 1
    assert max(words.values()) < 64 and len(set(words.values())) == len(words)     # TM_GS_DW, no word twice
    spec = importlib.util.spec_from_file_location("tm_store_py", os.path.join(ROOT, "tetris_mcts_amd", "store.py"))
    src = open(spec.origin).read()
    gs = eval("dict(" + src[src.index("GS = dict(") + 10:src.index(")", src.index("GS = dict("))] + ")")
    alias = {"CYC_GC16": ("CYC_TAIL", 0), "GC_REACHABLE": ("CYC_TAIL", 1)}
    for name, idx in gs.items():
        base, off = alias.get(name, (name, 0))
        assert words[base] + off == idx, (name, idx, words.get(base))
    for name in ("GC_PHASE", "GC_IN_MOVE", "GC_REQ_AT", "POOL_FULL", "SIM_TARGET", "SIM_STARTED"):
        assert name in gs
    L = _lib()
    assert L.TmStore.gc_spec_nodes.offset == L.TmStore.dist_bins.offset + 4


def test_hot_kernels_use_no_scratch_memory(tmp_path):
    """The kernels of the simulation step keep their working set in registers: a private segment (scratch) in one of them is a
    silent slowdown (r04: a 16-byte copy into a register array sent the array to scratch memory and fc1 from 35 to 53 us while
    every result stayed right).  Read from the code objects inside the built HIP objects."""
    import shutil
    import subprocess
    import pytest
    objdump, readelf = "/opt/rocm/lib/llvm/bin/llvm-objdump", "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not (os.path.exists(objdump) and os.path.exists(readelf)):
        pytest.skip("no llvm binutils")
    hot = ("k_sim_step", "k_vn_conv", "k_vn_fc1", "k_dn_conv", "k_dn_fc", "k_update_root", "k_root_stats")
    seen = set()
    for src in ("tree", "valuenet", "distnet"):
        obj = os.path.join(ROOT, "tetris_mcts_amd", "csrc", "_obj", src + ".o")
        if not os.path.exists(obj):
            pytest.skip("HIP objects not built")
        local = str(tmp_path / (src + ".o"))
        shutil.copy(obj, local)
        subprocess.check_call([objdump, "--offloading", local], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        cos = [f for f in os.listdir(tmp_path) if f.startswith(src + ".o.") and "amdgcn" in f]
        assert cos, "no device code object in " + obj
        notes = subprocess.check_output([readelf, "--notes", str(tmp_path / cos[0])]).decode()
        for name, scratch, spills in re.findall(r"\.name:\s+(\S+)\s.*?\.private_segment_fixed_size:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", notes, re.S):
            if any(h in name for h in hot):
                seen.add(next(h for h in hot if h in name))
                assert int(scratch) == 0 and int(spills) == 0, (name, scratch, spills)
    assert seen == set(hot), seen


def test_dense_request_list_addressing():
    """The dense request list of include/tetris_mcts_hip.h (tm_store::eval_list): game g appends to segment g % segs, entry d
    of segment s lives at (s + segs * (d / slots)) * slots + d % slots; the consumers (valuenet.hip req_prefix / req_at) count
    dense positions through the segments in order.  A model of both sides: every posted request is found exactly once, inside
    the slots of its segment's games, whatever the numbers of games, slots and requests."""
    rng = np.random.default_rng(7)
    for n_games, slots in ((1, 1), (3, 7), (64, 1), (100, 7), (4096, 1), (4096, 7), (777, 7)):
        segs = min(n_games, 64)
        posted = [int(rng.integers(0, slots + 1)) if rng.random() < 0.7 else 0 for _ in range(n_games)]
        cnt = [0] * segs
        lst = {}
        for g in rng.permutation(n_games):                       # the waves finish in any order
            k = posted[g]
            if k == 0:
                continue
            sg = g % segs
            base = cnt[sg]
            cnt[sg] += k
            for i in range(k):
                d = base + i
                at = (sg + segs * (d // slots)) * slots + d % slots
                assert 0 <= at < n_games * slots and at not in lst, (n_games, slots, g, at)
                assert (at // slots) % segs == sg                 # inside the slots of one of this segment's games
                lst[at] = (int(g) * slots + i)
        incl = np.cumsum(cnt)
        total = int(incl[-1])
        assert total == sum(posted)
        seen = set()
        for p in range(total):
            sg = int(np.sum(incl <= p))
            d = p - (int(incl[sg - 1]) if sg else 0)
            at = (sg + segs * (d // slots)) * slots + d % slots
            seen.add(lst[at])
        assert len(seen) == total


def test_a_slice_begins_at_a_multiple_of_four_games():
    """tm_store_slice (host arithmetic only: callable without a GPU): games [first, first + n) as a store of their own - `first` a
    multiple of four (a tree-kernel workgroup's games; the groups of the collectors' summary words, TM_GS_GC_ACTIVE4), anything
    else is refused; every per-game array advances by first x its per-game size."""
    L = _lib()
    lib = L.lib()
    s, out = L.TmStore(), L.TmStore()
    s.n_games, s.max_nodes, s.table_cap, s.eval_slots, s.max_trace, s.replay_cap = 64, 1000, 2048, 7, 256, 0
    for first, n, rc_ok in ((0, 64, True), (4, 60, True), (32, 8, True), (2, 8, False), (5, 4, False), (60, 8, False), (-4, 4, False)):
        rc = lib.tm_store_slice(C.byref(s), first, n, C.byref(out))
        assert (rc == 0) == rc_ok, (first, n, rc)
        if rc_ok:
            assert out.n_games == n
            gs0 = C.cast(s.gs, C.c_void_p).value or 0
            gs1 = C.cast(out.gs, C.c_void_p).value or 0
            assert gs1 - gs0 == first * 64 * 4          # the control blocks: 64 words a game
