"""Engine oracle self-checks against ENGINE_SPEC.md (parity with pyTetris itself is unpinned: its source is
absent from the reference tree)."""
import numpy as np
from hypothesis import given, settings, strategies as st


def test_bag_randomizer_is_a_permutation_per_bag(oracle):
    L = oracle.lib()
    for seed in (0, 1, 99, 2**32 - 1):
        seq = [L.orc_piece_at(seed, i, 0) for i in range(70)]
        for k in range(10):
            assert sorted(seq[7 * k:7 * k + 7]) == list(range(7))
        u = [L.orc_piece_at(seed, i, 1) for i in range(700)]
        assert set(u) == set(range(7))


def test_spawn_and_render(oracle):
    g = oracle.Game(seed=1)
    s = g.getState()
    assert (s == -1).sum() == 4 and (s == 1).sum() == 0
    assert set(np.argwhere(s == -1)[:, 0]) <= {0, 1}
    assert g.g["piece_count"][0] == 1 and g.combo == -1 and not g.end


@settings(max_examples=60, deadline=None)
@given(seed=st.integers(0, 2**32 - 1), app=st.integers(1, 3), scoring=st.integers(0, 1), rnd=st.integers(0, 1),
       actions=st.lists(st.integers(0, 6), min_size=1, max_size=400))
def test_invariants_under_random_play(oracle, seed, app, scoring, rnd, actions):
    g = oracle.Game(app, scoring, rnd, seed)
    prev_key = None
    for a in actions:
        before = g.g.copy()
        g.play(a)
        s = g.getState()
        rows = g.g["rows"][0]
        assert (rows < 0x3FF).all() or g.end  # no full row survives a lock
        assert (s == 1).sum() == sum(bin(int(r)).count("1") for r in rows)
        if g.end:
            assert (s == -1).sum() == 0
            again = g.g.copy()
            g.play(a)
            assert g.g.tobytes() == again.tobytes()  # play on an ended game is a no-op
            break
        assert (s == -1).sum() == 4
        # DAG property (ENGINE_SPEC 4): (piece_count, y, drop_ctr) strictly increases
        key = (int(g.g["piece_count"][0]), int(g.g["y"][0]), int(g.g["drop_ctr"][0]))
        if prev_key is not None and key[0] == prev_key[0]:
            assert key > prev_key
        prev_key = key
        assert g.g["line_clears"][0] == g.line_stats @ np.array([1, 2, 3, 4])
        assert g.score >= before["score"][0]
        # packed observation <-> rendered observation
        o = g.packed_obs()
        out = np.zeros((20, 10), np.int8)
        oracle.lib().orc_obs_render(oracle.ptr(o), oracle.ptr(out))
        assert np.array_equal(out, s)


def test_line_clear_scoring(oracle):
    # fill the bottom row except columns 3..6 by hand, then drop a flat I into the gap
    for seed in range(200):
        g = oracle.Game(seed=seed)
        if g.g["piece"][0] == 0:
            break
    g.g["rows"][0][19] = 0x3FF & ~(0xF << 3)
    g.play(3)
    assert g.line_clears == 1 and g.line_stats[0] == 1 and g.combo == 0
    assert g.score == 2 * 18 + 100  # 18 rows of hard drop + single
    assert (g.g["rows"][0] == 0).all()


# ---- directed cases (ENGINE_SPEC.md sections 4 and 5), the oracle against the specification's own numbers ----
def _game_with(oracle, rows, piece, rot=0, x=3, y=0, seed=9, **fields):
    g = oracle.Game(seed=seed)
    g.g["rows"][0] = rows
    g.g["piece"], g.g["rot"], g.g["x"], g.g["y"] = piece, rot, x, y
    for k, v in fields.items():
        g.g[k] = v
    return g


def test_single_double_triple_tetris_and_back_to_back(oracle):
    """Vertical I (orientation 1 occupies box column 2) dropped into a one-column well at column 9: n rows clear for n nearly
    full rows; base points 100/300/500/800, 1200 for a 4-line clear right after a 4-line clear (ENGINE_SPEC 5)."""
    base = {1: 100, 2: 300, 3: 500, 4: 800}
    for n in (1, 2, 3, 4):
        rows = np.zeros(20, np.uint16)
        rows[20 - n:] = 0x3FF & ~(1 << 9)
        g = _game_with(oracle, rows, piece=0, rot=1, x=7, y=0)
        g.play(3)
        dist = 16                                 # the I's lowest cell goes from row 3 to row 19
        assert g.line_clears == n and g.line_stats[n - 1] == 1 and g.combo == 0
        assert g.score == 2 * dist + base[n], (n, g.score)
        left = bin(int(np.bitwise_or.reduce(g.g["rows"][0]))).count("1")
        assert (left == 1) == (n < 4) and g.g["flags"][0] & 2 == (2 if n == 4 else 0)
    # back to back: two 4-line clears in a row, then a single ends the streak
    rows = np.zeros(20, np.uint16)
    rows[16:] = 0x3FF & ~(1 << 9)
    g = _game_with(oracle, rows, piece=0, rot=1, x=7, y=0)
    g.play(3)
    s1 = g.score
    g.g["rows"][0][16:] = 0x3FF & ~(1 << 9)
    g.g["piece"], g.g["rot"], g.g["x"], g.g["y"] = 0, 1, 7, 0
    g.play(3)
    assert g.score - s1 == 2 * 16 + 1200 + 50 * 1 and g.combo == 1          # back-to-back 1200, combo 1
    s2 = g.score
    g.g["rows"][0][19] = 0x3FF & ~(1 << 9)
    g.g["piece"], g.g["rot"], g.g["x"], g.g["y"] = 0, 1, 7, 0
    g.play(3)
    assert g.score - s2 == 2 * 16 + 100 + 50 * 2 and g.g["flags"][0] & 2 == 0 and g.combo == 2


def test_combo_resets_on_a_lock_without_clear(oracle):
    g = _game_with(oracle, np.zeros(20, np.uint16), piece=1, combo=3)
    g.play(3)
    assert g.combo == -1 and g.line_clears == 0


def test_top_out_at_spawn(oracle):
    rows = np.zeros(20, np.uint16)
    rows[0:2] = 0b0001111000            # the spawn area (columns 3-6, rows 0-1) is occupied
    rows[2:] = 0x3FE
    g = _game_with(oracle, rows, piece=1, rot=0, x=0, y=0)       # O piece locks at the top left -> next spawn collides
    g.g["rows"][0][0:2] = 0b0001111000
    g.play(3)
    assert g.end and (g.getState() == -1).sum() == 0
    before = g.g.copy()
    g.play(2)
    assert g.g.tobytes() == before.tobytes()


def test_wall_and_floor_contact_and_kicks(oracle):
    from engine_cases import piece_cells, valid
    cells = piece_cells(oracle)
    empty = np.zeros(20, np.uint16)
    # left / right against the walls do nothing but the gravity step
    g = _game_with(oracle, empty, piece=2, rot=0, x=0, y=5)
    g.play(1)
    assert (g.g["x"][0], g.g["y"][0]) == (0, 6)
    g = _game_with(oracle, empty, piece=2, rot=0, x=7, y=5)
    g.play(2)
    assert (g.g["x"][0], g.g["y"][0]) == (7, 6)
    # a vertical I against the left wall rotates with a kick to the right (I 1>2: (-1,0)(2,0)...: first valid), never upward
    g = _game_with(oracle, empty, piece=0, rot=1, x=-2, y=5)
    g.play(5)
    assert g.g["rot"][0] == 2 and valid(cells[(0, 2)], empty, int(g.g["x"][0]), int(g.g["y"][0]) - 1)
    # on the floor every kick that would move the piece up is skipped: rotation fails, the piece locks by gravity
    g = _game_with(oracle, empty, piece=0, rot=0, x=3, y=18)
    pc = int(g.g["piece_count"][0])
    g.play(5)
    assert int(g.g["piece_count"][0]) in (pc, pc + 1)
    # app = 3: gravity only every third action, a blocked soft drop locks at once
    g = oracle.Game(3, 0, 0, 4)
    y0 = int(g.g["y"][0])
    g.play(0); g.play(0)
    assert int(g.g["y"][0]) == y0
    g.play(0)
    assert int(g.g["y"][0]) == y0 + 1 and int(g.g["drop_ctr"][0]) == 0


def test_sweep_states_are_valid_and_cover_every_piece(oracle):
    from engine_cases import sweep_states
    recs = sweep_states(oracle, 2)
    assert len(recs) > 20000
    assert set(np.unique(recs["piece"])) == set(range(7)) and set(np.unique(recs["rot"])) == set(range(4))
    # every action on every state keeps the invariants (no full row, 4 falling cells or ended)
    L = oracle.lib()
    ls = np.zeros(4, np.int32)
    for i in range(0, len(recs), 97):
        for a in range(8):
            g = recs[i:i + 1].copy()
            L.orc_game_play(oracle.ptr(g), 2, 0, 0, a, oracle.ptr(ls))
            assert (g["rows"][0] < 0x3FF).all()
