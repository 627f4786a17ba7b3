"""Per-wave phase cycles of k_sim_step at the benchmark's workload (4096 games x 500 sims, pool 100 000, the bench's weights and
seeds): after every move the control words of every game's LAST simulation (backup / select / verification / expansion cycles,
trace length, verified prefix, requests posted) -> gpurun_out/wave_cycles.npz.  Input of scripts/rowp_model.py."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tetris_mcts_amd import agents, store as st, dist as tdist  # noqa: E402
from tetris_mcts_amd.model import Model_VV  # noqa: E402
from tetris_mcts_amd.pyTetris import Tetris  # noqa: E402

G = int(os.environ.get("G", "4096"))
moves = int(os.environ.get("MOVES", "25"))
out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/wave_cycles.npz"
env_args = ((20, 10), 1, 0, 0)
model = Model_VV(backend="hip", seed=0)
game = Tetris(*env_args, seed=tdist.game_seeds(20260925, G, 0), n_games=G)
agent = agents.ValueSim(sims=500, env=Tetris, env_args=env_args, n_games=G, max_nodes=100000, model=model, online=False)
agent.update_root(game)
words = ("CYC_BACK", "CYC_SELECT", "CYC_VERIFY", "CYC_EXPAND", "TRACE_LEN", "FIRST_MISS", "K_EVAL", "LEAF_END")
rec = {k: [] for k in words}
for m in range(moves):
    a = agent.play()
    gs = agent.store.t["gs"].cpu().numpy()
    for k in words:
        rec[k].append(gs[:, st.GS[k]].copy())
    game.play(a)
    agent.update_root(game)
    if np.atleast_1d(game.end).any():
        game.reset("ended")
        agent.update_root(game)
np.savez_compressed(out, **{k: np.stack(v) for k, v in rec.items()})
tot = np.stack(rec["CYC_BACK"])[5:] + np.stack(rec["CYC_SELECT"])[5:] + np.stack(rec["CYC_EXPAND"])[5:]
print("wave cycles (moves 6..): mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f  trace mean %.1f max %d" % (
    tot.mean(), np.percentile(tot, 50), np.percentile(tot, 90), np.percentile(tot, 99), tot.max(),
    np.stack(rec["TRACE_LEN"])[5:].mean(), np.stack(rec["TRACE_LEN"])[5:].max()))
