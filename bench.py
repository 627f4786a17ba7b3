#!/usr/bin/env python
"""bench.py — the headline benchmark of BASELINE.json on MI355X.

Metric: MCTS node-expansions/sec at 4096 concurrent games x 500 simulations/move (BASELINE configs[1]:
ValueSim UCT with the value-net leaf evaluator), one process per GPU, games sharded (weak scaling).
A "step" is one move of every game: 500 simulations (select / expand / evaluate / backup) for each of the
rank's games, then get_action, game.play and update_root.  State is resident in HBM before the timed region.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

np = None

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_STATE = 3803136            # conv1 82,944 + conv2 1,769,472 + conv3 1,032,192 + fc1 917,504 + fc_out 1,024
PEAK_F32_MATRIX_TFLOPS = 157.3      # MI355X_MICROARCH.md: FP32 matrix peak (dense)
PEAK_HBM_GBPS = 8000.0              # MI355X_MICROARCH.md: HBM3E peak


def pmc_traffic(kernels, fetch_scale=1.0):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/r01_pmc_traffic.json; counters can
    not be read from inside this process, and the guide asks for separate passes).  FETCH_SIZE + WRITE_SIZE are
    KiB per dispatch; fetch_scale = 2 for kernels whose reads are wide coalesced streams (gfx950 under-report)."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        k = json.load(f)["kernels"]
    tot = 0.0
    for name in kernels:
        e = k.get(name)
        if not e:
            return None
        tot += 1024.0 * (fetch_scale * e["FETCH_SIZE_KB_last50_mean"] + e["WRITE_SIZE_KB_last50_mean"])
    return tot


def bytes_per_sim(mean_trace_len, k_eval):
    """SURVEY.md 8(d) algorithmic bytes per simulation (packed game 64 B, packed observation 64 B, U = 7)."""
    return 204.0 * mean_trace_len - 144.0 + 1296.0 + 1000.0 * k_eval


def main():
    global np
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)     # 5 warm-up moves, then 20 timed moves
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--games", type=int, default=4096, help="games per GPU")
    ap.add_argument("--sims", type=int, default=500)
    ap.add_argument("--agent", default="ValueSim", choices=["ValueSim", "ValueSimLP"])
    ap.add_argument("--max-nodes", type=int, default=100000)
    ap.add_argument("--backend", default="hip", choices=["hip", "torch"])
    ap.add_argument("--graph", action="store_true", help="replay the simulation body as a hipGraph (no per-kernel events)")
    ap.add_argument("--split", type=int, default=1, help="run the rank's games as this many independent sub-batches on "
                    "separate HIP streams, so one sub-batch's tree kernel overlaps another's value-net kernels")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    assert world == args.gpus, "launch with --nproc-per-node equal to --gpus"

    import __graft_entry__ as ge
    if rank == 0 and not os.path.exists(ge.LIB):
        ge.build()
    if world > 1:
        dist.barrier()
    from tetris_mcts_amd import agents, store as st
    from tetris_mcts_amd.model import Model_VV
    from tetris_mcts_amd.pyTetris import Tetris

    G, sims = args.games, args.sims
    NS = args.split
    assert G % NS == 0
    Gs = G // NS
    env_args = ((20, 10), 1, 0, 0)
    model = Model_VV(backend=args.backend, seed=0)  # model_vv.Net() under torch.manual_seed(0) (random init)
    base_seed = 20260925 + rank * G
    streams = [torch.cuda.Stream() for _ in range(NS)] if NS > 1 else [torch.cuda.current_stream()]
    games, agts, models = [], [], []
    for k in range(NS):
        with torch.cuda.stream(streams[k]):
            mk = model if k == 0 else Model_VV(backend=args.backend, seed=0)   # own scratch per stream, same weights
            game = Tetris(*env_args, seed=base_seed + k * Gs, n_games=Gs)
            agent = getattr(agents, args.agent)(sims=sims, env=Tetris, env_args=env_args, n_games=Gs,
                                                max_nodes=args.max_nodes, model=mk, online=False, use_graph=args.graph)
            agent.update_root(game)
        games.append(game); agts.append(agent); models.append(mk)
    torch.cuda.synchronize()
    stores = [a.store for a in agts]
    dev = stores[0].device
    K = stores[0].eval_slots

    ev = [[[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(sims)] for _ in range(NS)] if not args.graph else None
    # HIP events bracket every EV_EVERY-th simulation of the timed region (two event records per launch cost ~7 us of
    # stream time per simulation, 3 % of a step, when placed around all of them)
    EV_EVERY = int(os.environ.get("TM_BENCH_EVENT_EVERY", "16"))
    t_tree = t_nn = 0.0
    n_tree = n_nn = 0
    episodes, lines = 0, 0

    def one_step(timed):
        nonlocal t_tree, t_nn, n_tree, n_nn, episodes, lines
        if args.graph:
            for k in range(NS):
                with torch.cuda.stream(streams[k]):
                    agts[k].mcts(sims)
        else:
            for i in range(sims):
                for k in range(NS):
                    with torch.cuda.stream(streams[k]):
                        e = ev[k][i] if i % EV_EVERY == 0 else None
                        if e: e[0].record()
                        stores[k].sim_step(st.SIM_BACKUP | st.SIM_FRONT)
                        if e: e[1].record()
                        agts[k].evaluate_requests()
                        if e: e[2].record()
            for k in range(NS):
                with torch.cuda.stream(streams[k]):
                    stores[k].sim_step(st.SIM_BACKUP)
        for k in range(NS):
            with torch.cuda.stream(streams[k]):
                _, action = stores[k].root_stats()
                games[k].play(action)
                agts[k].update_root(games[k])   # reads game.end: one small host sync per move, as the reference's loop has
                ended = np.atleast_1d(games[k].end)
                if ended.any():
                    episodes += int(ended.sum())
                    lines += int(np.atleast_1d(games[k].line_clears)[ended].sum())
                    games[k].reset("ended")
                    agts[k].update_root(games[k])
        if timed and not args.graph:
            torch.cuda.synchronize()
            for k in range(NS):
                for e in ev[k][::EV_EVERY]:
                    t_tree += e[0].elapsed_time(e[1])
                    t_nn += e[1].elapsed_time(e[2])
            n_tree += len(ev[0][::EV_EVERY]) * NS
            n_nn += len(ev[0][::EV_EVERY]) * NS

    def counters():
        return {k: sum(S.counter(k) for S in stores) for k in ("N_EXPAND", "N_SIMS", "TRACE_SUM", "N_EVAL")}

    for _ in range(args.warmup):
        one_step(False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    c0 = counters()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step(True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    c1 = counters()
    err = sum(int((S.errors() != 0).sum().item()) for S in stores)
    d = {k: c1[k] - c0[k] for k in c0}
    tot = torch.tensor([elapsed, d["N_EXPAND"], d["N_SIMS"], d["TRACE_SUM"], d["N_EVAL"], episodes, lines, err],
                       dtype=torch.float64, device=dev)
    if world > 1:
        tmax = tot[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        tot[0] = tmax[0]
    elapsed, n_exp, n_sims, tr_sum, n_eval, episodes, lines, err = [float(x) for x in tot.cpu()]

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    mean_len = tr_sum / max(n_sims, 1.0)
    out = {
        "metric": "mcts_node_expansions_per_sec",
        "value": n_exp / elapsed,
        "unit": "node-expansions/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": "%d games/GPU x %d sims/move, %s UCT with value-net leaf evaluation (BASELINE configs[%d]); "
                        "Tetris 20x10 app=1 guideline scoring 7-bag; node pool %d/game; Net() random init manual_seed(0)"
                        % (G, sims, args.agent, 1 if args.agent == "ValueSim" else 2, args.max_nodes),
            "games_per_gpu": G, "sims_per_move": sims, "agent": args.agent, "max_nodes": args.max_nodes,
            "valuenet_backend": args.backend, "graph": bool(args.graph), "streams": NS,
        },
        "sims_per_sec": n_sims / elapsed,
        "child_steps_per_sec": 7.0 * n_exp / elapsed,
        "evaluated_states_per_sec": n_eval / elapsed,
        "mean_trace_len": mean_len,
        "episodes_finished": int(episodes),
        "lines_cleared_per_episode": (lines / episodes) if episodes else None,
        "error_games": int(err),
        "store_gib_per_gpu": sum(S.nbytes() for S in stores) / 2**30,
        "last_sim_phase_kcycles": {k: float(np.mean([S.t["gs"][:, st.GS[k]].float().mean().item() for S in stores])) / 1e3
                                   for k in ("CYC_BACK", "CYC_SELECT", "CYC_EXPAND", "TRACE_LEN")},
    }
    if not args.graph and n_nn:
        nn_ms, tree_ms = t_nn / n_nn, t_tree / n_tree
        evals_per_launch = n_eval / (args.steps * sims * world * NS)
        flops = FLOP_PER_STATE * evals_per_launch
        a_tf = flops / (nn_ms * 1e-3) / 1e12
        bps = bytes_per_sim(mean_len, 1 if args.agent == "ValueSim" else n_eval / max(n_exp, 1.0))
        a_gbs = bps * Gs / (tree_ms * 1e-3) / 1e9
        nn_roof = {"kernel": "value net forward (tm_valuenet_forward + eval render), per launch of %d states" % (Gs * K),
                   "bound": "mfma", "achieved": a_tf, "peak": PEAK_F32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                   "frac": a_tf / PEAK_F32_MATRIX_TFLOPS,
                   "traffic": pmc_traffic(["tmcts_vn::k_vn_conv", "tmcts_vn::k_vn_fc1", "tmcts_vn::k_fc_out"], 2.0)
                   if args.backend == "hip" else None,
                   "traffic_note": "bytes/launch, FETCH_SIZE x2 (wide streams) + WRITE_SIZE from profiles/r01_pmc_traffic.json",
                   "avg_launch_ms": nn_ms, "launches_timed": int(n_nn), "events_every": EV_EVERY}
        tree_roof = {"kernel": "k_sim_step (backup+select+expand), per launch of %d games" % Gs, "bound": "hbm",
                     "achieved": a_gbs, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": a_gbs / PEAK_HBM_GBPS,
                     "traffic": pmc_traffic(["tmcts::k_sim_step<false>"]),
                     "traffic_note": "bytes/launch of 4096 games at mean trace length ~20 (move 1), FETCH_SIZE + WRITE_SIZE "
                                     "from profiles/r01_pmc_traffic.json (narrow scattered accesses: no gfx950 correction)",
                     "avg_launch_ms": tree_ms, "algorithmic_bytes_per_sim": bps, "launches_timed": int(n_tree),
                     "events_every": EV_EVERY}
        out["roofline"] = nn_roof if nn_ms >= tree_ms else tree_roof
        out["roofline_other"] = tree_roof if nn_ms >= tree_ms else nn_roof
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, model)
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(args, model):
    """CPU baseline on this box's host cores, 1 thread, bounded sample of the same workload (1 game, same sims/move,
    same network weights).  Preferred: kind "reference" = the reference's own all-C++ MCTSAgent (agents/cppmodule/
    agent.cpp compiled in place into oracle/_ref/, the ValueSimC path) with a single-thread torch CPU evaluator.
    Fallback / secondary: kind "port" = the oracle's C restatement incl. its fma-chain value net."""
    import torch
    from oracle import binding as B
    out = None
    lp = args.agent != "ValueSim"
    try:
        agent_mod = B.load_ref_native("agent")
        if agent_mod is not None:
            sys.path.insert(0, B.BUILD)
            from pyTetris import Tetris as OTetris
            from tetris_mcts_amd.model import Net
            torch.set_num_threads(1)
            net = Net().eval()
            net.load_state_dict({k: v.detach().cpu() for k, v in model.model.state_dict().items()})
            n_calls = [0, 0]

            def ev(obs):
                x = torch.from_numpy(np.asarray(obs).astype(np.float32))
                with torch.no_grad():
                    y = net(x.reshape(-1, 1, 20, 10))
                n_calls[0] += 1
                n_calls[1] += y.shape[0]
                if lp:
                    return [y[:, 0].tolist(), y[:, 1].tolist()]
                return [float(y[0, 0]), float(y[0, 1])]
            import ctypes
            ctypes.CDLL("libc.so.6").srand(1)
            ag = agent_mod.MCTSAgent(args.sims, args.max_nodes, True, 0.999, False, ev, 0, lp)
            g = OTetris((20, 10), 1, 0, 0, 20260925)
            ag.update_root(g)
            t0 = time.perf_counter()
            moves = 0
            while time.perf_counter() - t0 < args.cpu_seconds:
                g.play(int(ag.play()))
                ag.update_root(g)
                moves += 1
                if g.end:
                    g.reset()
                    ag.update_root(g)
            dt = time.perf_counter() - t0
            out = {"value": n_calls[0] / dt, "unit": "node-expansions/s", "cores": 1, "kind": "reference",
                   "sample": "1 game x %d moves x %d sims/move, the reference's own MCTSAgent (agent.cpp compiled in place, "
                             "LP=%s) + torch CPU Net, 1 thread, %.1f s" % (moves, args.sims, lp, dt),
                   "sims_per_sec": moves * args.sims / dt, "evaluated_states_per_sec": n_calls[1] / dt,
                   "host_cpus": os.cpu_count()}
    except Exception as e:  # the reference build is optional on the GPU box
        out = None
        sys.stderr.write("reference CPU baseline unavailable (%s); using the oracle port\n" % (e,))
    kind = 0 if args.agent == "ValueSim" else 1
    params = model.flat_params().cpu().numpy()
    g = B.Game(seed=20260925)
    a = B.Agent(kind, max_nodes=args.max_nodes, evaluator="valuenet", params=params)
    a.update_root(g)
    t0 = time.perf_counter()
    moves = 0
    budget = args.cpu_seconds if out is None else min(args.cpu_seconds, 5.0)
    while time.perf_counter() - t0 < budget:
        g.play(a.play(args.sims))
        a.update_root(g)
        moves += 1
        if g.end:
            g.reset()
            a.update_root(g)
    dt = time.perf_counter() - t0
    port = {"value": a.n_expand / dt, "unit": "node-expansions/s", "cores": 1, "kind": "port",
            "sample": "1 game x %d moves x %d sims/move, oracle C restatement of %s incl. fp32 value net, %.1f s" %
                      (moves, args.sims, args.agent, dt),
            "sims_per_sec": a.n_sims / dt, "host_cpus": os.cpu_count()}
    if out is None:
        return port
    out["port"] = port
    return out


if __name__ == "__main__":
    main()
