"""How much of a walk repeats the previous walk of the same game?  Runs the benchmark's search for a number of moves and
prints, from the per-game control blocks, the length of the last walk and the level at which it first left its nodes'
predicted children (TM_GS_FIRST_MISS), overall and for the longest walks (the ones a launch waits for).

    python scripts/walk_prefix.py [--moves 12] [--games 4096] [--sims 500]"""
import argparse, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tetris_mcts_amd import agents
from tetris_mcts_amd.model import Model_VV
from tetris_mcts_amd.pyTetris import Tetris

ap = argparse.ArgumentParser()
ap.add_argument("--moves", type=int, default=12)
ap.add_argument("--games", type=int, default=4096)
ap.add_argument("--sims", type=int, default=500)
ap.add_argument("--agent", default="ValueSim")
a = ap.parse_args()
env_args = ((20, 10), 1, 0, 0)
model = Model_VV(backend="hip", seed=0)
game = Tetris(*env_args, seed=20260925, n_games=a.games)
agent = getattr(agents, a.agent)(sims=a.sims, env=Tetris, env_args=env_args, n_games=a.games, max_nodes=100000, model=model, online=False)
agent.update_root(game)
out = []
for m in range(a.moves):
    act = agent.play()
    gs = agent.store.t["gs"].cpu().numpy()
    ln, fm = gs[:, 4].astype(np.int64), gs[:, 28].astype(np.int64)
    order = np.argsort(-ln)
    top = order[:41]
    rec = dict(move=m, mean_len=float(ln.mean()), mean_first_miss=float(fm.mean()), max_len=int(ln.max()),
               first_miss_of_longest=int(fm[order[0]]),
               top1pct_len=float(ln[top].mean()), top1pct_first_miss=float(fm[top].mean()),
               prefix_sum_over_trace_sum=float(gs[:, 29].sum() / max(gs[:, 16].sum(), 1)),
               frac_first_miss_ge_90pct=float((fm >= 0.9 * ln - 1).mean()),
               top1pct_remaining_levels=float((ln[top] - fm[top]).mean()), mean_remaining_levels=float((ln - fm).mean()),
               max_remaining_levels=int((ln - fm).max()))
    cyc = {k: gs[:, i].astype(np.int64) for k, i in (("back", 20), ("select", 21), ("expand", 22), ("verify", 42))}
    tot = cyc["back"] + cyc["select"] + cyc["expand"]
    slow = np.argsort(-tot)[:8]
    rec["kcycles_mean"] = {k: float(v.mean()) / 1e3 for k, v in cyc.items()}
    rec["kcycles_p99"] = {k: float(np.percentile(v, 99)) / 1e3 for k, v in cyc.items()}
    rec["kcycles_total_mean_p99_max"] = [float(tot.mean()) / 1e3, float(np.percentile(tot, 99)) / 1e3, float(tot.max()) / 1e3]
    rec["slowest_games"] = [dict(g=int(g), len=int(ln[g]), k0=int(fm[g]), pool_full=int(gs[g, 44]), pending=int(gs[g, 5]), nfree=int(gs[g, 2]),
                                 gc_phase=int(gs[g, 32] & 15), **{k: int(v[g]) // 1000 for k, v in cyc.items()}) for g in slow]
    rec["pool_full_games"] = int((gs[:, 44] != 0).sum()); rec["collecting_games"] = int(((gs[:, 32] & 15) != 0).sum())
    out.append(rec)
    if m >= a.moves - 3: print(json.dumps(rec), flush=True)
    game.play(act)
    agent.update_root(game)
    ended = np.atleast_1d(game.end)
    if ended.any():
        game.reset("ended")
        agent.update_root(game)
