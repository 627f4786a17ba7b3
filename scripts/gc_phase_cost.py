"""Diagnostic (build with -DTM_GC_TIMING): shader cycles the first bounded workgroup spends per step of a collection."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tetris_mcts_amd import agents
from tetris_mcts_amd.model import Model_VV
from tetris_mcts_amd.pyTetris import Tetris
env_args = ((20, 10), 1, 0, 0)
model = Model_VV(backend="hip", seed=0)
G = 4096
game = Tetris(*env_args, seed=20260925, n_games=G)
agent = agents.ValueSim(sims=500, env=Tetris, env_args=env_args, n_games=G, max_nodes=100000, model=model, online=False)
agent.update_root(game)
for m in range(int(sys.argv[1]) if len(sys.argv) > 1 else 84):
    act = agent.play(); game.play(act); agent.update_root(game)
    ended = np.atleast_1d(game.end)
    if ended.any(): game.reset("ended"); agent.update_root(game)
gs = agent.store.t["gs"].cpu().numpy()
sel = gs[:, 9] > 0
names = {1: "init", 2: "mark (sum over launches)", 3: "count", 4: "write", 5: "nodes", 6: "obs"}
for ph, nm in names.items():
    v = gs[sel, 48 + ph].astype(np.float64) * 16 / 2400.0      # microseconds at 2.4 GHz
    print("%-26s mean %7.1f us  p50 %7.1f  p90 %7.1f  max %7.1f   (games %d)" % (nm, v.mean(), np.percentile(v, 50), np.percentile(v, 90), v.max(), sel.sum()))
print("reachable at last GC: mean", gs[sel, 24].mean())
