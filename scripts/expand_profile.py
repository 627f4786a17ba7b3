"""Instrumented build only (scripts/build_variant.sh profexp -DTM_PROF_EXPAND; TETRIS_MCTS_LIB=variants/libtetris_profexp.so):
mean cycles since the start of the expansion at each probe (control words 48..57), 4096 games, after `moves` moves."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from tetris_mcts_amd import agents  # noqa: E402
from tetris_mcts_amd.model import Model_VV  # noqa: E402
from tetris_mcts_amd.pyTetris import Tetris  # noqa: E402

G, sims, moves = 4096, 500, int(sys.argv[1]) if len(sys.argv) > 1 else 10
env_args = ((20, 10), 1, 0, 0)
game = Tetris(*env_args, seed=20260925, n_games=G)
agent = agents.ValueSim(sims=sims, env=Tetris, env_args=env_args, n_games=G, max_nodes=100000, model=Model_VV(backend="hip", seed=0))
agent.update_root(game)
for _ in range(moves):
    a = agent.play()
    game.play(a)
    agent.update_root(game)
torch.cuda.synchronize()
gs = agent.store.t["gs"][:, 48:58].float().cpu().numpy()
names = ["load+replicate", "play x7", "nn: dedupe+hash", "nn: node table find", "nn: node alloc+insert", "nn: obs pack+dedupe",
         "nn: obs table find", "nn: obs alloc+insert", "new_nodes done (writes)", "unique children + record"]
prev = 0.0
for i, n in enumerate(names):
    m = float(gs[:, i].mean())
    print("%-28s at %8.0f  (+%6.0f)" % (n, m, m - prev))
    prev = m
