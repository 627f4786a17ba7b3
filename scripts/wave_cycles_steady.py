"""Per-wave phase cycles of k_sim_step in the steady state (pools full, collections under way), optionally under a checkpoint:
after each of the last moves the control words of every game's LAST simulation.  What the launch's slowest waves are made of.
    python scripts/wave_cycles_steady.py [--checkpoint F] [--warm-moves 80] [--moves 6]"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tetris_mcts_amd import agents, store as st, dist as tdist  # noqa: E402
from tetris_mcts_amd.model import Model_VV  # noqa: E402
from tetris_mcts_amd.pyTetris import Tetris  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--checkpoint", default=None); ap.add_argument("--warm-moves", type=int, default=80); ap.add_argument("--moves", type=int, default=6)
ap.add_argument("--games", type=int, default=4096)
a = ap.parse_args()
G = a.games
env_args = ((20, 10), 1, 0, 0)
model = Model_VV(backend="hip", seed=0)
if a.checkpoint: model.load(a.checkpoint, verbose=False)
game = Tetris(*env_args, seed=tdist.game_seeds(20260925, G, 0), n_games=G)
agent = agents.ValueSim(sims=500, env=Tetris, env_args=env_args, n_games=G, max_nodes=100000, model=model, online=False)
agent.update_root(game)
for m in range(a.warm_moves + a.moves):
    act = agent.play()
    if m >= a.warm_moves:
        gs = agent.store.t["gs"].cpu().numpy()
        b, s_, v, e = (gs[:, st.GS[k]].astype(np.float64) for k in ("CYC_BACK", "CYC_SELECT", "CYC_VERIFY", "CYC_EXPAND"))
        tot = b + s_ + e
        free, full, tl = gs[:, st.GS["NFREE_NODE"]], gs[:, st.GS["POOL_FULL"]], gs[:, st.GS["TRACE_LEN"]]
        pc = lambda x: "mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f" % (x.mean(), np.percentile(x, 50), np.percentile(x, 90), np.percentile(x, 99), x.max())
        print("move %d: wave kcycles of the last simulation: total [%s]" % (m, pc(tot / 1e3)))
        print("    backup [%s]  select [%s]  expand [%s]  trace mean %.1f max %d" % (pc(b / 1e3), pc(s_ / 1e3), pc(e / 1e3), tl.mean(), tl.max()))
        slow = np.argsort(-tot)[:6]
        print("    slowest: " + "; ".join("g%d %.0fk (b %.0f s %.0f e %.0f) free %d full %d len %d" % (g, tot[g] / 1e3, b[g] / 1e3, s_[g] / 1e3, e[g] / 1e3, free[g], full[g], tl[g]) for g in slow))
        lowfree = free < 256
        print("    games with < 256 free nodes: %d, their expand mean %.0fk against %.0fk; pool-full games %d expand mean %.0fk" % (
            lowfree.sum(), e[lowfree].mean() / 1e3 if lowfree.any() else 0, e[~lowfree].mean() / 1e3, (full != 0).sum(), e[full != 0].mean() / 1e3 if (full != 0).any() else 0), flush=True)
    game.play(act)
    agent.update_root(game)
    if np.atleast_1d(game.end).any():
        game.reset("ended")
        agent.update_root(game)
