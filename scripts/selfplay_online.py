"""Online self-play learning curve: batched ValueSimLP with GC-harvested TD targets and periodic fits of the value net
(the reference's `python play.py --agent_type ValueSimLP --online ...`, README.md:36, scaled to many games at once), or
`--agent DistValueSim`: the distributional head fitted on the distributions of the nodes the collections free
(DistValueSimOnline.py:116-170).  Writes one JSON line per training round: the episodes finished since the last round with
their mean lines / score, and - without survivorship bias - the lines cleared per 1000 moves by ALL games in the round and the
mean over every episode that was under way in it (finished ones with their final count, the others with their count so far)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tetris_mcts_amd import agents, model as M  # noqa: E402
from tetris_mcts_amd.pyTetris import Tetris  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--games", type=int, default=512)
ap.add_argument("--sims", type=int, default=200)
ap.add_argument("--agent", default="ValueSimLP")
ap.add_argument("--max-nodes", type=int, default=30000)
ap.add_argument("--minutes", type=float, default=8.0)
ap.add_argument("--train-every", type=int, default=100, help="moves between training attempts")
ap.add_argument("--train-iters", type=int, default=3000)
ap.add_argument("--min-visits", type=int, default=None, help="min_visits_to_store (default: the agent's)")
ap.add_argument("--out", default="gpurun_out/online_learning.jsonl")
ap.add_argument("--save", default=None, help="checkpoint file (the reference's format, model/model.py:152-160), written every --save-every rounds and at the end")
ap.add_argument("--save-every", type=int, default=10)
ap.add_argument("--load", default=None, help="start from this checkpoint instead of a random-init net")
ap.add_argument("--no-train", action="store_true", help="play only (with --load: the checkpoint's play strength): no harvest, no fits")
args = ap.parse_args()

M.EXP_PATH = "/tmp/tm_ckpt/"
os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
G = args.games
env_args = ((20, 10), 1, 0, 0)
game = Tetris(*env_args, seed=1234, n_games=G)
extra = {} if args.min_visits is None else dict(min_visits_to_store=args.min_visits)
if args.agent.startswith("Dist"):
    from tetris_mcts_amd.model_distributional import Model_Dist
    model = Model_Dist(atoms=50, seed=0, backend="hip")
else:
    model = M.Model_VV(backend="hip", seed=0)
if args.load:
    assert os.path.isfile(args.load), args.load
    model.load(args.load, verbose=False)
agent = getattr(agents, args.agent)(sims=args.sims, env=Tetris, env_args=env_args, n_games=G, max_nodes=args.max_nodes,
                                    model=model, online=not args.no_train, replay_cap=0 if args.no_train else 16384, **extra)
agent.update_root(game)
t0 = t_round = time.time()
train_total = 0.0
moves, rounds = 0, 0
ep_lines, ep_scores, ep_len = [], [], []
alive = np.zeros(G, np.int64)
lines_round, moves_round = 0, 0          # lines cleared / moves played by all games since the last round
prev_lines = np.zeros(G, np.int64)
log = open(args.out, "w")
while time.time() - t0 < args.minutes * 60:
    act = agent.play()
    game.play(act)
    agent.update_root(game)
    moves += 1
    alive += 1
    cur = np.asarray(game.line_clears, np.int64)
    lines_round += int((cur - prev_lines).sum())
    moves_round += G
    prev_lines = cur.copy()
    ended = game.end
    if ended.any():
        ep_lines += list(game.line_clears[ended])
        ep_scores += list(game.score[ended])
        ep_len += list(alive[ended])
        alive[ended] = 0
        prev_lines[ended] = 0
        game.reset("ended")
        agent.update_root(game)
    if moves % args.train_every == 0:
        tuples = int(agent.store.t["replay_count"].sum().item())
        tt = time.time()
        res = None if args.no_train else agent.train_nodes(iters_per_val=100, batch_size=1024, max_iters=args.train_iters, log=False)
        rounds += 1
        train_total += time.time() - tt
        rec = dict(round=rounds, t=round(time.time() - t0, 1), moves=moves, episodes=len(ep_lines),
                   mean_lines=float(np.mean(ep_lines)) if ep_lines else None,
                   max_lines=int(np.max(ep_lines)) if ep_lines else None,
                   mean_score=float(np.mean(ep_scores)) if ep_scores else None,
                   mean_episode_moves=float(np.mean(ep_len)) if ep_len else None,
                   lines_per_1000_moves=1000.0 * lines_round / max(moves_round, 1),
                   mean_lines_all_episodes_under_way=float(np.mean(list(ep_lines) + list(prev_lines))),
                   new_tuples=tuples, trained=res is not None, train_iters=(res or {}).get("iters"),
                   best_val=(res or {}).get("best_validation"), train_s=round(time.time() - tt, 2),
                   train_ms_per_iter=(1e3 * (time.time() - tt) / res["iters"]) if res and res.get("iters") else None,
                   train_share_of_wall_so_far=round(train_total / max(time.time() - t0, 1e-9), 3),
                   gcs=agent.store.counter("N_GC"), gc_slices=agent.store.counter("GC_SLICES"),
                   dropped_tuples=agent.store.counter("N_DROPPED"), pool_resets=agent.store.counter("N_POOL_RESET"),
                   **{k: int((agent.store.search_stats(agent.n_sub, agent.ev_every, reset=False) or {}).get(k, 0))
                      for k in ("tree_launches", "catchup_launches", "gc_launches")},
                   round_s=round(time.time() - t_round, 1))
        t_round = time.time()
        log.write(json.dumps(rec) + "\n")
        log.flush()
        print(rec, flush=True)
        ep_lines, ep_scores, ep_len = [], [], []
        lines_round, moves_round = 0, 0
        if args.save and rounds % args.save_every == 0:
            model.save(filename=args.save, verbose=False)
log.close()
if args.save:
    model.save(filename=args.save, verbose=False)
    print("checkpoint written to", args.save, flush=True)
