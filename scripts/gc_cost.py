"""Cost of one in-kernel GC (gc_wave) as a function of pool size and reachable set."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from gpu_helpers import hash_eval_torch
from tetris_mcts_amd import agents, store as st
from tetris_mcts_amd.pyTetris import Tetris
CASES = {"small": ((20000, 100),), "large": ((100000, 200),), "all": ((20000, 100), (100000, 200))}
for mn, sims in CASES[sys.argv[1] if len(sys.argv) > 1 else "all"]:
    G = 256
    game = Tetris((20, 10), 1, 0, 0, seed=7, n_games=G)
    agent = agents.ValueSimLP(sims=sims, env=Tetris, env_args=game.env_args, n_games=G, max_nodes=mn, evaluator=hash_eval_torch)
    agent.update_root(game)
    for m in range(400):
        act = agent.play(); game.play(act); agent.update_root(game)
        if game.end.any():
            game.reset("ended"); agent.update_root(game)
        gs = agent.store.t["gs"].cpu().numpy()
        if (gs[:, 9] > 0).sum() > G // 2:
            break
    sel = gs[:, 9] > 0
    print("pool", mn, "moves", m + 1, "games with GC", int(sel.sum()), "mean GC kcycles", round(float(gs[sel, 23].mean()) * 16 / 1e3, 1),
          "= %.2f ms @2.1GHz" % (float(gs[sel, 23].mean()) * 16 / 2.1e6), "reachable mean", int(gs[sel, 24].mean()), flush=True)
