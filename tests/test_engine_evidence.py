"""The only records of the real pyTetris engine that the reference tree holds: demo/test.gif (200 frames of
a 22 x 10 game) and results/online-200sims/log_endless (per-episode score / lines).  scripts/reference_evidence.py decodes
them into tests/golden/ref_engine_evidence.npz; here the report is re-derived from that file and held against
ENGINE_SPEC.md and the oracle engine's own tables."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _load():
    import reference_evidence as E
    d = np.load(os.path.join(ROOT, "tests", "golden", "ref_engine_evidence.npz"))
    return E, d["gif_boards"], d["log_endless"]


def test_gif_every_piece_shape_lock_and_clear_follow_the_specification():
    E, boards, log = _load()
    rep = E.analyse(boards, log)
    assert rep["frames"] == 200 and rep["unknown_piece_shapes"] == 0            # only the 7 tetrominoes, in ENGINE_SPEC 3's orientations
    assert rep["piece_types_seen"] == ["I", "J", "L", "O", "S", "T", "Z"]
    assert rep["never_climbs"] == rep["same_piece_transitions"] == 150         # a piece only descends (no upward kicks seen)
    # every change of the locked cells = the previous piece, at rest, locked and full rows removed (ENGINE_SPEC 5)
    assert rep["lock_events"] == rep["lock_events_explained_by_one_resting_piece_plus_line_clear"] == 49
    assert rep["line_clears_seen"]["1"] == 9 and rep["line_clears_seen"]["2"] == 6
    # a new piece is first seen in rows 1-3 of the 22-row board (= rows -1..1 of today's 20), columns 2-7: the spawn area
    assert set(rep["first_seen_after_spawn_top_row_22"]) <= {1, 2, 3}
    assert all(2 <= a and b <= 7 for a, b in rep["first_seen_after_spawn_cols"])


def test_gif_shapes_are_the_oracle_engines_orientation_tables(oracle):
    """The shapes in the frames, normalised to their bounding box, against what the ORACLE engine renders for each
    (piece, orientation) - the table the HIP engine is tested against bit for bit."""
    from engine_cases import piece_cells
    E, boards, _ = _load()
    cells = piece_cells(oracle)
    ours = {E.norm([(dy, dx) for dx, dy in cl]) for cl in cells.values()}
    seen = set()
    for b in boards:
        fall = [(int(r), int(c)) for r, c in np.argwhere(b == -1)]
        if len(fall) == 4:
            seen.add(E.norm(fall))
    assert seen <= ours and len(seen) >= 15, (len(seen), len(ours))            # 19 distinct shapes exist; most occur in 200 frames


def test_log_endless_scores_fit_the_scoring_table():
    """log_endless: zero-line episodes score 262-486 (drop points only: 2 per hard-dropped row, 1 per soft-dropped row in
    ENGINE_SPEC 5) and the score grows by ~200 per cleared line (100 for a single + the drop points of the ~2.5 pieces a
    line takes + combos).  The same agent (ValueSim, 200 simulations, untrained network) on the oracle engine scores
    530-600 with one line and 740-980 with three to five (ENGINE_SPEC.md section 9 records that run)."""
    E, _, log = _load()
    rep = E.analyse(np.zeros((0, 22, 10), np.int8), log)["log_endless"]
    assert rep["episodes"] == 524 and rep["zero_line_episodes"] == 19
    lo, mean, hi = rep["zero_line_score_min_mean_max"]
    assert 200 < lo and hi < 600 and 150 < rep["score_per_line_slope"] < 260
