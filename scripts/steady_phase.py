"""Diagnostic: per-game phase cycles of a REGULAR launch in the steady state (pools full), Python-driven for a few launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tetris_mcts_amd import agents, store as st
from tetris_mcts_amd.model import Model_VV
from tetris_mcts_amd.pyTetris import Tetris
env_args = ((20, 10), 1, 0, 0)
model = Model_VV(backend="hip", seed=0)
G = 4096
warm = int(sys.argv[1]) if len(sys.argv) > 1 else 84
game = Tetris(*env_args, seed=20260925, n_games=G)
agent = agents.ValueSim(sims=500, env=Tetris, env_args=env_args, n_games=G, max_nodes=100000, model=model, online=False)
agent.update_root(game)
s = agent.store
for m in range(warm):
    act = agent.play(); game.play(act); agent.update_root(game)
    ended = np.atleast_1d(game.end)
    if ended.any(): game.reset("ended"); agent.update_root(game)
both = st.SIM_BACKUP | st.SIM_FRONT
s.move_begin(500); s.sim_step(both)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(120):
    agent.evaluate_requests()
    ev0.record(); s.sim_step(both); ev1.record()
    if i % 40 == 39:
        torch.cuda.synchronize()
        gs = s.t["gs"].cpu().numpy()
        cyc = {k: gs[:, j].astype(np.int64) for k, j in (("back", 20), ("select", 21), ("expand", 22), ("verify", 42))}
        tot = cyc["back"] + cyc["select"] + cyc["expand"]
        act_ = (gs[:, 5] == 1)
        print("launch", i, "kernel ms", round(ev0.elapsed_time(ev1), 4), "active games", int(act_.sum()), "collecting", int(((gs[:, 32] & 15) != 0).sum()), "pool_full", int((gs[:, 44] != 0).sum()))
        print("  mean kcycles (active):", {k: round(float(v[act_].mean()) / 1e3, 1) for k, v in cyc.items()}, "total mean/p99/max", round(float(tot[act_].mean()) / 1e3, 1), round(float(np.percentile(tot[act_], 99)) / 1e3, 1), round(float(tot.max()) / 1e3, 1))
        slow = np.argsort(-tot)[:6]
        for g in slow:
            print("   g", int(g), "len", int(gs[g, 4]), "k0", int(gs[g, 28]), "pool_full", int(gs[g, 44]), "nfree", int(gs[g, 2]), {k: int(v[g]) // 1000 for k, v in cyc.items()})
