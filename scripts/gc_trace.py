"""Diagnostic: the launch-by-launch course of a garbage collection (phase word, arrivals, queue length) of the first
games that collect, with the search driven launch by launch from Python.
    python scripts/gc_trace.py [--games 256] [--max-nodes 20000] [--sims 300] [--moves 40]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tetris_mcts_amd import agents, store as st
from tetris_mcts_amd.model import Model_VV
from tetris_mcts_amd.pyTetris import Tetris
ap = argparse.ArgumentParser()
ap.add_argument("--games", type=int, default=256); ap.add_argument("--max-nodes", type=int, default=20000)
ap.add_argument("--sims", type=int, default=300); ap.add_argument("--moves", type=int, default=40)
ap.add_argument("--slice", type=int, default=110000); ap.add_argument("--warm-moves", type=int, default=0)
a = ap.parse_args()
env_args = ((20, 10), 1, 0, 0)
model = Model_VV(backend="hip", seed=0)
game = Tetris(*env_args, seed=20260925, n_games=a.games)
agent = agents.ValueSim(sims=a.sims, env=Tetris, env_args=env_args, n_games=a.games, max_nodes=a.max_nodes, model=model, online=False, gc_slice_cycles=a.slice)
agent.update_root(game)
s = agent.store
both = st.SIM_BACKUP | st.SIM_FRONT
watch, log, launch = None, [], 0
hist = []
def finish_move():
    act = agent.get_action(); game.play(act); agent.update_root(game)
    ended = np.atleast_1d(game.end)
    if ended.any(): game.reset("ended"); agent.update_root(game)
for m in range(a.warm_moves):
    s.search(a.sims, model); finish_move()
print("warm moves done; collections so far", s.counter("N_GC"), flush=True)
import time
for m in range(a.warm_moves, a.warm_moves + a.moves):
    tm0 = time.perf_counter(); l0 = launch
    s.move_begin(a.sims); s.sim_step(both); launch += 1
    todo = a.sims
    while todo > 0:
        for _ in range(todo):
            agent.evaluate_requests(); s.sim_step(both); launch += 1
            gs = s.t["gs"][:, 32:36].cpu().numpy()
            col = np.nonzero((gs[:, 0] & 15) != 0)[0]
            if hist is not None and len(hist) < 120:
                hist.append((launch, np.bincount(gs[:, 0] & 15, minlength=8).tolist()))
            if watch is None and len(col): watch = int(col[0])
            if watch is not None and len(log) < 400:
                w = gs[watch]
                log.append((launch, m, int(w[0] & 15), int(w[0] >> 4), int(w[1] & 255), int(w[1] >> 8), int(w[2]), int(w[3]), len(col)))
        todo, collecting = s.sims_remaining()
        while collecting:
            for _ in range(6): s.gc_step(); launch += 1
            todo, collecting = s.sims_remaining()
    finish_move()
    print("move", m, "launches", launch - l0, "collections", s.counter("N_GC"), "collecting now", int(((s.t["gs"][:, 32] & 15) != 0).sum().item()), flush=True)
    if watch is not None and len(log) >= 400: break
for h in hist: print("launch", h[0], "games by phase [none, req, mark, count, write, nodes, obs, done]", h[1])
print("game", watch)
prev = None
for r in log:
    key = r[2:4] + r[6:8]
    if key != prev: print("launch %d move %d phase %d seq %d arrive %d left %d tail %d tail0 %d collecting %d" % r)
    prev = key
