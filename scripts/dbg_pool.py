import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tetris_mcts_amd import agents, model as M, store as st
from tetris_mcts_amd.pyTetris import Tetris
M.EXP_PATH = "/tmp/tm_ckpt/"
G = 512
env_args = ((20, 10), 1, 0, 0)
game = Tetris(*env_args, seed=1234, n_games=G)
model = M.Model_VV(backend="hip", seed=0)
agent = agents.ValueSimLP(sims=200, env=Tetris, env_args=env_args, n_games=G, max_nodes=30000, model=model, online=True, replay_cap=8192)
agent.update_root(game)
for m in range(80):
    agent.mcts(agent.sims)
    gs = agent.store.t["gs"].cpu().numpy()
    bad = np.nonzero(gs[:, 6])[0]
    if len(bad):
        print("move", m, "bad games", bad[:10], "err", gs[bad[:10], 6])
        for g in bad[:3]:
            print(" g", g, {k: int(gs[g, v]) for k, v in st.GS.items()})
        print(" nfree node min/mean", gs[:, 2].min(), gs[:, 2].mean(), "nfree obs min", gs[:, 3].min(), "gc total", gs[:, 9].sum(), "trace max", gs[:, 4].max())
        break
    _, action = agent.store.root_stats()
    game.play(action)
    agent.update_root(game)
    if game.end.any():
        game.reset("ended"); agent.update_root(game)
else:
    gs = agent.store.t["gs"].cpu().numpy()
    print("no error; nfree min", gs[:, 2].min(), "gc", gs[:, 9].sum())
