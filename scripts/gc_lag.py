"""Diagnostic: which games are behind at the end of a move's regular launches, and why (collections in the move,
free nodes, reachable nodes at the last collection, phase), the search driven launch by launch from Python after a
native warm-up.
    python scripts/gc_lag.py [--games 4096] [--max-nodes 100000] [--sims 500] [--warm-moves 75] [--moves 6]"""
import argparse, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tetris_mcts_amd import agents, store as st
from tetris_mcts_amd.model import Model_VV
from tetris_mcts_amd.pyTetris import Tetris
ap = argparse.ArgumentParser()
ap.add_argument("--games", type=int, default=4096); ap.add_argument("--max-nodes", type=int, default=100000)
ap.add_argument("--sims", type=int, default=500); ap.add_argument("--moves", type=int, default=6)
ap.add_argument("--warm-moves", type=int, default=75); ap.add_argument("--gc-spec-nodes", type=int, default=None)
ap.add_argument("--checkpoint", default=None, help="value-net checkpoint (the trained regime)")
ap.add_argument("--native-moves", type=int, default=0, help="after the traced moves: this many moves through the native loop, counters only")
a = ap.parse_args()
env_args = ((20, 10), 1, 0, 0)
model = Model_VV(backend="hip", seed=0)
if a.checkpoint: model.load(a.checkpoint, verbose=False)
game = Tetris(*env_args, seed=20260925 + np.arange(a.games, dtype=np.int64), n_games=a.games)
agent = agents.ValueSim(sims=a.sims, env=Tetris, env_args=env_args, n_games=a.games, max_nodes=a.max_nodes, model=model,
                        online=False, gc_spec_nodes=a.gc_spec_nodes)
agent.update_root(game)
s = agent.store
both = st.SIM_BACKUP | st.SIM_FRONT
def finish_move():
    act = agent.get_action(); game.play(act); agent.update_root(game)
    ended = np.atleast_1d(game.end)
    if ended.any(): game.reset("ended"); agent.update_root(game)
for m in range(a.warm_moves):
    s.search(a.sims, model); finish_move()
print("warm moves done; collections so far", int(s.t["gs"][:, 9].sum().item()), "pool", a.max_nodes, flush=True)
for m in range(a.moves):
    g0 = s.t["gs"].cpu().numpy().copy()
    s.move_begin(a.sims); s.sim_step(both)
    hist = torch.zeros(a.games, 16, dtype=torch.int32, device=s.device)      # launches spent in each phase
    ar = torch.arange(a.games, device=s.device)
    for i in range(a.sims):
        agent.evaluate_requests(); s.sim_step(both)
        hist[ar, (s.t["gs"][:, 32] & 15).long()] += 1
    hist = hist.cpu().numpy()
    spec = hist[:, 8] + hist[:, 9]
    stall = hist[:, 1:8].sum(axis=1) + hist[:, 10:].sum(axis=1)
    g1 = s.t["gs"].cpu().numpy()
    lag = g1[:, 40] - g1[:, 41] + (g1[:, 5] != 0)
    ngc = g1[:, 9] - g0[:, 9]
    print("move %d: lag max %d, games behind %d; lag histogram (0,1-5,6-10,11-20,21-40,41-80,>80): %s; collections %d in %d games, max per game %d"
          % (m, lag.max(), int((lag > 1).sum()), np.histogram(lag, bins=[0, 1, 6, 11, 21, 41, 81, 10 ** 6])[0].tolist(), ngc.sum(), int((ngc > 0).sum()), ngc.max()), flush=True)
    for g in np.argsort(-lag)[:8]:
        print("   game %4d lag %3d collections %d stalled launches %3d speculative launches %3d free nodes %5d reachable at last gc %5d phase %2d pool_full %d slices %d"
              % (g, lag[g], ngc[g], stall[g], spec[g], g1[g, 2], g1[g, 24], g1[g, 32] & 15, g1[g, 44], g1[g, 38] - g0[g, 38]))
        print("        launches by phase [req, mark, count, write, nodes, obs, done | spec-req, spec-mark, req-spec, -, req-over]:", hist[g, 1:8].tolist(), hist[g, 8:13].tolist())
        ml = max(1, g1[g, 37] - g0[g, 37])
        print("        marker: launches %d, workgroups per launch %.2f, games its workgroups had %.2f, blocks %d, rounds %d, kcycles of its workgroups per launch %.0f, wave kcycles per block %.1f, idle turns %d"
              % (g1[g, 37] - g0[g, 37], (g1[g, 58] - g0[g, 58]) / ml, (g1[g, 59] - g0[g, 59]) / ml, g1[g, 51] - g0[g, 51], g1[g, 52] - g0[g, 52],
                 (g1[g, 53] - g0[g, 53]) * 64 / 1e3 / ml, (g1[g, 62] - g0[g, 62]) * 64 / 1e3 / max(1, g1[g, 51] - g0[g, 51]), g1[g, 63] - g0[g, 63]))
    col = ngc >= 1
    if col.any():
        names = ["req", "mark", "count", "write", "nodes", "obs", "done", "spec-req", "spec-mark", "req-spec", "-", "req-over"]
        print("   games that collected (%d): launches by phase, mean per collection: %s" % (int(col.sum()),
              ", ".join("%s %.1f" % (n, hist[col, 1 + i].sum() / ngc[col].sum()) for i, n in enumerate(names))))
        first = np.array([np.argmax(hist[g, 1:8] > 0) if hist[g, 1:8].any() else -1 for g in np.nonzero(col)[0]])
        print("   marker per collection in this move: launches %.1f, blocks %.0f, workgroups per launch %.2f, games per workgroup %.2f" % (
              (g1[col, 37] - g0[col, 37]).sum() / ngc[col].sum(), (g1[col, 51] - g0[col, 51]).sum() / ngc[col].sum(),
              (g1[col, 58] - g0[col, 58]).sum() / max(1, (g1[col, 37] - g0[col, 37]).sum()),
              (g1[col, 59] - g0[col, 59]).sum() / max(1, (g1[col, 37] - g0[col, 37]).sum())))
    one = ngc == 1
    if one.any():
        print("   games with one collection: stalled launches mean %.1f max %d; speculative launches mean %.1f" % (stall[one].mean(), stall[one].max(), spec[one].mean()))
    todo, collecting = s.sims_remaining()
    n_catch = 0
    while todo > 0 or collecting:
        while collecting:
            for _ in range(6): s.gc_step()
            todo, collecting = s.sims_remaining()
        for _ in range(todo):
            agent.evaluate_requests(); s.sim_step(both); n_catch += 1
        todo, collecting = s.sims_remaining()
    print("   catch-up launches", n_catch, flush=True)
    finish_move()

for m in range(a.native_moves):
    g0 = s.t["gs"].cpu().numpy().copy()
    s.search_stats(1, 0, reset=True)
    s.search(a.sims, model)
    ss = s.search_stats(1, 0, reset=False)
    g1 = s.t["gs"].cpu().numpy()
    ngc = (g1[:, 9] - g0[:, 9]).sum()
    print("native move %d: collections %d, waiting launches per collection %.1f, marker launches per collection %.1f, catch-up launches %d, collector-only launches %d"
          % (m, ngc, (g1[:, 38] - g0[:, 38]).sum() / max(1, ngc), (g1[:, 37] - g0[:, 37]).sum() / max(1, ngc), ss["catchup_launches"], ss["gc_launches"]), flush=True)
    finish_move()
