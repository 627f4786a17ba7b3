"""HIP engine (tetris_mcts_amd/csrc/engine.h) vs the CPU oracle engine: bit-exact packed games."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("app,scoring,rnd", [(1, 0, 0), (1, 1, 1), (2, 0, 1), (3, 0, 0)])
def test_env_step_matches_oracle(oracle, app, scoring, rnd):
    import torch
    from tetris_mcts_amd.pyTetris import Tetris
    G, steps = 192, 400
    rng = np.random.default_rng(app * 100 + scoring * 10 + rnd)
    env = Tetris((20, 10), app, scoring, rnd, seed=777, n_games=G)
    ora = [oracle.Game(app, scoring, rnd, 777 + g) for g in range(G)]
    assert env.packed().tobytes() == b"".join(o.g.tobytes() for o in ora)
    for t in range(steps):
        # bias towards drops so games finish, lines clear and resets happen
        a = rng.choice(7, size=G, p=[0.1, 0.15, 0.15, 0.25, 0.1, 0.125, 0.125]).astype(np.int32)
        env.play(a)
        for g in range(G):
            ora[g].play(int(a[g]))
        if t % 7 == 0 or t == steps - 1:
            got = env.packed()
            exp = np.concatenate([o.g for o in ora]).view(np.uint32).reshape(G, 16)
            bad = np.nonzero((got != exp).any(axis=1))[0]
            assert len(bad) == 0, (t, bad[:5], got[bad[0]], exp[bad[0]])
            assert np.array_equal(env.line_stats, np.stack([o.line_stats for o in ora]))
            st = env.getState()
            for g in range(0, G, 37):
                assert np.array_equal(st[g], ora[g].getState())
            ended = np.array([o.end for o in ora])
            assert np.array_equal(env.end, ended)
            if ended.any():
                env.reset("ended")
                for g in np.nonzero(ended)[0]:
                    ora[g].reset()
    assert sum(o.line_clears for o in ora) >= 0
    torch.cuda.synchronize()


@pytest.mark.parametrize("app,scoring", [(1, 0), (2, 1), (3, 0)])
def test_state_injection_sweep(oracle, app, scoring):
    """Every piece x orientation x valid position (resting, one above, near the top) on crafted boards (1-4 nearly full
    rows with wells, stairs, walls, a high shaft, random junk), every drop counter, both back-to-back states: each state
    is injected into the device games and stepped with each of the actions 0..7, HIP vs oracle, whole packed game and
    line statistics.  Covers singles to tetrises, back-to-back, combo counting, wall / floor kicks, the skipped upward
    kicks, locks by gravity and by a blocked soft drop, top-outs at spawn."""
    import torch
    from engine_cases import sweep_states
    from tetris_mcts_amd.pyTetris import Tetris
    recs = sweep_states(oracle, app)
    n = len(recs)
    L = oracle.lib()
    env = Tetris((20, 10), app, scoring, 0, seed=1, n_games=n)
    cleared = np.zeros(5, np.int64)
    for a in range(8):
        env.games.copy_(torch.from_numpy(recs.view(np.int32).reshape(n, 16).copy()))
        env._ls.zero_()
        env.play(np.full(n, a, np.int32))
        got = env.packed()
        got_ls = env._ls.cpu().numpy()
        exp = recs.copy()
        exp_ls = np.zeros((n, 4), np.int32)
        for i in range(n):
            L.orc_game_play(oracle.ptr(exp[i:i + 1]), app, scoring, 0, a, oracle.ptr(exp_ls[i]))
        expw = exp.view(np.uint32).reshape(n, 16)
        bad = np.nonzero((got != expw).any(axis=1))[0]
        assert len(bad) == 0, (a, len(bad), recs[bad[0]], got[bad[0]], expw[bad[0]])
        assert np.array_equal(got_ls, exp_ls)
        cleared[1:] += exp_ls.sum(axis=0)
    assert (cleared[1:] > 0).all(), cleared         # singles, doubles, triples and tetrises all occurred
