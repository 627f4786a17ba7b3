"""When the workgroups of ONE k_sim_step launch start and end (ticks of s_memrealtime, its rate measured against the host clock between two sampled moves), in the steady state: the last regular launch (over all games) of each of the last moves - every simulation wave and every collector workgroup.  Needs a library built with
-DTM_TIMELINE (scripts/build_variant.sh timeline tree.hip '1i #define TM_TIMELINE 1'), which stamps words of the game blocks.
    TETRIS_MCTS_LIB=$PWD/build_variants/timeline.so python scripts/launch_timeline.py [--checkpoint F] [--warm-moves 80] [--moves 6] [--out F.json]"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tetris_mcts_amd import agents, store as st, dist as tdist  # noqa: E402
from tetris_mcts_amd.model import Model_VV  # noqa: E402
from tetris_mcts_amd.pyTetris import Tetris  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--checkpoint", default=None); ap.add_argument("--warm-moves", type=int, default=80); ap.add_argument("--moves", type=int, default=6)
ap.add_argument("--warm-sample", type=int, default=1, help="sampled moves before the reported ones (they calibrate the counter)")
ap.add_argument("--games", type=int, default=4096); ap.add_argument("--collectors", type=int, default=128); ap.add_argument("--out", default=None)
a = ap.parse_args()
G = a.games
env_args = ((20, 10), 1, 0, 0)
model = Model_VV(backend="hip", seed=0)
if a.checkpoint: model.load(a.checkpoint, verbose=False)
game = Tetris(*env_args, seed=tdist.game_seeds(20260925, G, 0), n_games=G)
agent = agents.ValueSim(sims=500, env=Tetris, env_args=env_args, n_games=G, max_nodes=100000, model=model, online=False)
agent.update_root(game)
pc = lambda x: {"mean": round(float(x.mean()), 2), "p50": round(float(np.percentile(x, 50)), 2), "p90": round(float(np.percentile(x, 90)), 2),
                "p99": round(float(np.percentile(x, 99)), 2), "max": round(float(x.max()), 2)} if len(x) else None
rows = []
prev = None      # (tick of the previous sampled launch, host time after its move): the counter's rate, measured
tick_hz = []
for m in range(a.warm_moves + a.moves):
    act = agent.play()
    if m >= a.warm_moves - a.warm_sample:
        gs = agent.store.t["gs"].cpu().numpy()
        u = lambda k: gs[:, st.GS[k] if isinstance(k, str) else k].astype(np.uint32).astype(np.int64)
        launch = u("CYC_EXPAND")
        vals, cnt = np.unique(launch, return_counts=True)
        the = vals[np.argmax(cnt)]                # the move's last launch that starts a simulation for every game
        sel = launch == the
        s0, s1 = u("CYC_BACK")[sel], u("CYC_SELECT")[sel]
        o, nc = 3 * a.collectors * int(the & 1), a.collectors       # (the collectors' stamps of even / odd launches are kept apart;
        c0, c1, cl = u(35)[o:o + nc], u(35)[o + nc:o + 2 * nc], u(35)[o + 2 * nc:o + 3 * nc]      #  the spare word 35: start, end, launch)
        if ((c1 >> 16) != (the & 0xFFFF)).any(): cl = cl * 0 - 1      # (the end words carry their launch too)
        c1 = c0 + (c1 & 0xFFFF)                    # end = start + ticks (kept relative: the 32-bit clock wraps every 43 s)
        if not (cl == the).all():
            print("move %d: the collectors' stamps are of launch %s, the simulation waves' of %d - skipped" % (m, np.unique(cl), the), flush=True)
            continue_ = True
        else: continue_ = False
        ref = min(s0.min(), c0.min())
        now = time.perf_counter()
        if prev is not None: tick_hz.append((ref - prev[0]) / (now - prev[1]))
        prev = (ref, now)
        hz = float(np.median(tick_hz)) if tick_hz else 100e6      # (the first sampled move is printed with the nominal 100 MHz)
        us = lambda x: (x - ref) / hz * 1e6
        nb = a.collectors // 2
        row = {"move": m, "launch": int(the), "collector_stamps_of_this_launch": not continue_, "simulating_waves": int(sel.sum()), "trace_len_mean": float(gs[sel, st.GS["TRACE_LEN"]].mean()),
               "launch_us": float(us(max(s1.max(), c1.max()))),
               "sim_start_us": pc(us(s0)), "sim_end_us": pc(us(s1)), "sim_duration_us": pc((s1 - s0) / hz * 1e6), "tick_mhz": round(hz / 1e6, 2),
               "collector_bounded_start_us": pc(us(c0[:nb])), "collector_bounded_end_us": pc(us(c1[:nb])),
               "collector_marking_start_us": pc(us(c0[nb:])), "collector_marking_end_us": pc(us(c1[nb:])),
               "last_sim_wave_end_us": float(us(s1.max())), "last_collector_end_us": float(us(c1.max()))}
        # which waves end last: their duration and start
        order = np.argsort(-s1)[:5]
        row["slowest_sim_waves"] = [{"start_us": float(us(s0[i])), "end_us": float(us(s1[i])), "trace_len": int(gs[sel, st.GS["TRACE_LEN"]][i])} for i in order]
        if m >= a.warm_moves:
            rows.append(row)
            print(json.dumps(row), flush=True)
    game.play(act)
    agent.update_root(game)
    if np.atleast_1d(game.end).any():
        game.reset("ended")
        agent.update_root(game)
if a.out:
    json.dump({"what": __doc__, "checkpoint": a.checkpoint, "rows": rows}, open(a.out, "w"), indent=1)
