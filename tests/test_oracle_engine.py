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
