"""What the reference tree itself records about the (absent) pyTetris engine, turned into checks of ENGINE_SPEC.md.

    python scripts/reference_evidence.py            (build container only: reads /root/reference)

1. demo/test.gif - 200 frames of a real game of an older pyTetris (22 x 10 board: the two rows that are hidden today are
   drawn, util/gui.py:7), sampled from `board_output` every 0.1 s (util/gui.py:30-39), so consecutive frames can be
   several actions apart.  Decoded to boards (0 empty, 1 locked, -1 falling; colours util/gui.py:15) and checked:
   piece shapes, spawn cells, monotone descent, and - by brute force over every placement of the previous piece - that
   each change of the locked cells is exactly ENGINE_SPEC section 5's lock + line clear of one piece at rest.
2. results/online-200sims/log_endless - per-episode (score, lines) of a training run: the score of zero-line episodes and
   the slope of score over lines against ENGINE_SPEC's scoring table.
Writes tests/golden/ref_engine_evidence.npz (decoded boards + parsed log) and prints the report that ENGINE_SPEC.md quotes;
tests/test_engine_evidence.py re-derives the report from that file (it travels; the reference does not)."""
import json
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("TETRIS_MCTS_REFERENCE", "/root/reference")

# ENGINE_SPEC.md section 3: cells (col, row) of every piece orientation inside its 4x4 box
CELLS = {
    "I": [[(0, 1), (1, 1), (2, 1), (3, 1)], [(2, 0), (2, 1), (2, 2), (2, 3)], [(0, 2), (1, 2), (2, 2), (3, 2)], [(1, 0), (1, 1), (1, 2), (1, 3)]],
    "O": [[(1, 0), (2, 0), (1, 1), (2, 1)]] * 4,
    "T": [[(1, 0), (0, 1), (1, 1), (2, 1)], [(1, 0), (1, 1), (2, 1), (1, 2)], [(0, 1), (1, 1), (2, 1), (1, 2)], [(1, 0), (0, 1), (1, 1), (1, 2)]],
    "S": [[(1, 0), (2, 0), (0, 1), (1, 1)], [(1, 0), (1, 1), (2, 1), (2, 2)], [(1, 1), (2, 1), (0, 2), (1, 2)], [(0, 0), (0, 1), (1, 1), (1, 2)]],
    "Z": [[(0, 0), (1, 0), (1, 1), (2, 1)], [(2, 0), (1, 1), (2, 1), (1, 2)], [(0, 1), (1, 1), (1, 2), (2, 2)], [(1, 0), (0, 1), (1, 1), (0, 2)]],
    "J": [[(0, 0), (0, 1), (1, 1), (2, 1)], [(1, 0), (2, 0), (1, 1), (1, 2)], [(0, 1), (1, 1), (2, 1), (2, 2)], [(1, 0), (1, 1), (0, 2), (1, 2)]],
    "L": [[(2, 0), (0, 1), (1, 1), (2, 1)], [(1, 0), (1, 1), (1, 2), (2, 2)], [(0, 1), (1, 1), (2, 1), (0, 2)], [(0, 0), (1, 0), (1, 1), (1, 2)]],
}


def norm(cells):
    """(row, col) cells -> shape key: sorted offsets from the bounding box corner."""
    r0, c0 = min(r for r, c in cells), min(c for r, c in cells)
    return tuple(sorted((r - r0, c - c0) for r, c in cells))


SHAPES = {}
for name, rots in CELLS.items():
    for k, cl in enumerate(rots):
        SHAPES.setdefault(norm([(r, c) for c, r in cl]), (name, k))


def decode_gif(path):
    from PIL import Image
    im = Image.open(path)
    out = []
    for i in range(im.n_frames):
        im.seek(i)
        f = np.array(im.convert("L"))
        b = np.zeros((22, 10), np.int8)
        for y in range(22):
            for x in range(10):
                b[y, x] = {0: 0, 255: 1, 122: -1}[int(f[10 + 20 * y + 10, 10 + 20 * x + 10])]   # 20 px cells, 10 px margin
        out.append(b)
    return np.stack(out)


def parse_log(path):
    rows = []
    for line in open(path, errors="replace"):
        m = re.match(r"Episode:\s*(\d+)\s+Score:\s*(-?\d+)\s+Lines Cleared:\s*(\d+)", line)
        if m:
            rows.append([int(m.group(1)), int(m.group(2)), int(m.group(3))])
    return np.asarray(rows, np.int64)


def lock_and_clear(locked, cells):
    """ENGINE_SPEC section 5 on a 22-row board: add the cells, remove full rows, shift the rows above down."""
    b = locked.copy()
    for r, c in cells:
        b[r, c] = 1
    keep = [r for r in range(b.shape[0]) if not b[r].all()]
    n = b.shape[0] - len(keep)
    out = np.zeros_like(b)
    out[n:] = b[keep]
    return out, n


def placements(locked, name):
    """Every resting placement of piece `name` on `locked` (any orientation, any column)."""
    H, W = locked.shape
    for k in range(4):
        cl = CELLS[name][k]
        for x in range(-2, W):
            for y in range(-2, H):
                cells = [(y + r, x + c) for c, r in cl]
                if any(r < 0 or r >= H or c < 0 or c >= W or locked[r, c] for r, c in cells):
                    continue
                below = [(r + 1, c) for r, c in cells]
                if any(r >= H or locked[r, c] for r, c in below if (r, c) not in cells):
                    yield k, x, y, cells


def analyse(boards, log):
    rep = dict(frames=int(len(boards)))
    unknown_shapes, spawn_rows, spawn_cols, pieces = 0, [], [], []
    descents_ok = moves = 0
    locks = explained = 0
    clears = {0: 0, 1: 0, 2: 0, 3: 0, 4: 0}
    prev = None
    for i, b in enumerate(boards):
        fall = [(int(r), int(c)) for r, c in np.argwhere(b == -1)]
        locked = (b == 1).astype(np.int8)
        name = None
        if len(fall) == 4:
            key = norm(fall)
            if key in SHAPES:
                name = SHAPES[key][0]
            else:
                unknown_shapes += 1
        if prev is not None:
            pfall, plocked, pname = prev
            if (locked != plocked).any():
                locks += 1
                ok = False
                if pname is not None:
                    for k, x, y, cells in placements(plocked, pname):
                        res, n = lock_and_clear(plocked, cells)
                        if (res == locked).all():
                            ok = True
                            clears[n] += 1
                            break
                explained += ok
                if len(fall) == 4:     # the next piece, at most a few actions after its spawn
                    spawn_rows.append(min(r for r, c in fall))
                    spawn_cols.append((min(c for r, c in fall), max(c for r, c in fall)))
                    pieces.append(name)
            elif len(fall) == 4 and len(pfall) == 4 and name == pname:
                moves += 1
                dy = min(r for r, c in fall) - min(r for r, c in pfall)
                descents_ok += dy >= -1      # a rotation can raise the bounding box by one row; the piece never climbs
        prev = (fall, locked, name)
    rep.update(unknown_piece_shapes=unknown_shapes, same_piece_transitions=moves, never_climbs=int(descents_ok),
               lock_events=locks, lock_events_explained_by_one_resting_piece_plus_line_clear=int(explained),
               line_clears_seen={str(k): v for k, v in clears.items()},
               first_seen_after_spawn_top_row_22=dict(zip(*[x.tolist() for x in np.unique(spawn_rows, return_counts=True)])),
               first_seen_after_spawn_cols=sorted(set(spawn_cols)), piece_types_seen=sorted(set(p for p in pieces if p)))
    if len(log):
        sc, ln = log[:, 1].astype(float), log[:, 2].astype(float)
        zero = sc[ln == 0]
        slope, icpt = np.polyfit(ln, sc, 1)
        hi = log[ln >= 20]
        rep["log_endless"] = dict(episodes=int(len(log)), zero_line_episodes=int(len(zero)),
                                  zero_line_score_min_mean_max=[float(zero.min()), float(zero.mean()), float(zero.max())] if len(zero) else None,
                                  score_per_line_slope=float(slope), intercept=float(icpt),
                                  score_per_line_in_episodes_with_20plus_lines=float((hi[:, 1] / hi[:, 2]).mean()) if len(hi) else None)
    return rep


if __name__ == "__main__":
    boards = decode_gif(os.path.join(REF, "demo", "test.gif"))
    log = parse_log(os.path.join(REF, "results", "online-200sims", "log_endless"))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_engine_evidence.npz"), gif_boards=boards, log_endless=log)
    print(json.dumps(analyse(boards, log), indent=1))
