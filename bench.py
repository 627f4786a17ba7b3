#!/usr/bin/env python
"""bench.py — the headline benchmark of BASELINE.json on MI355X.

Metric: MCTS node-expansions/sec at 4096 concurrent games x 500 simulations/move (BASELINE configs[1]:
ValueSim UCT with the value-net leaf evaluator), one process per GPU, games sharded (weak scaling).
A "step" is one move of every game: 500 simulations (select / expand / evaluate / backup) for each of the
rank's games through the native launch loop (search.hip), then get_action, game.play and update_root.
State is resident in HBM before the timed region.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
    ... --online     harvest (state, TD-target) tuples at every garbage collection and all-gather them over the ranks
                     after every move inside the timed region (BASELINE configs[3]: RCCL used for exactly that)
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
PMC_FILE = os.path.join("profiles", "r0[456]_pmc_traffic*.json")   # one file per profiled command line (workload_key); the latest round's wins


def pmc_traffic(kernels, workload_key, fetch_scale=1.0):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes of THIS workload (counters can not be read from
    inside this process, and the guide asks for separate passes): only used when the file was recorded with the same
    command line (its "workload_key"), otherwise the traffic is reported as null.  FETCH_SIZE + WRITE_SIZE are KiB per
    dispatch; fetch_scale = 2 for kernels whose reads are wide coalesced streams (gfx950 under-report)."""
    import glob
    doc = None
    for path in sorted(glob.glob(os.path.join(ROOT, PMC_FILE))):
        with open(path) as f:
            d = json.load(f)
        if d.get("workload_key") == workload_key:
            doc = d
    if doc is None:
        return None
    tot = 0.0
    for name in kernels:
        # (a kernel template is listed with its arguments, e.g. k_vn_fc1<1, 256, 6>: one instantiation runs per workload)
        es = [e for k, e in doc["kernels"].items() if k == name or k.startswith(name + "<")]
        if len(es) != 1 or "FETCH_SIZE_KB_mean" not in es[0] or "WRITE_SIZE_KB_mean" not in es[0]:
            return None
        tot += 1024.0 * (fetch_scale * es[0]["FETCH_SIZE_KB_mean"] + es[0]["WRITE_SIZE_KB_mean"])
    return tot


FLOP_PER_STATE_DIST = 2770432       # model_distributional.Net on 22x10: conv1 136,192 + conv2 2,097,152 + fc1 524,288 + fc_v 12,800


def bytes_per_sim_dist(mean_trace_len):
    """Algorithmic bytes per simulation of the distributional agent: per level 128 B record + 7 x 16 B
    child statistics (select), 16 B trace entry + 2 x 16 B statistics + 2 x 200 B distribution (backup); expansion and leaf
    gather as SURVEY.md 8(d) (1296 B; 200 B observation read + 880 B network input written)."""
    return (240.0 + 448.0) * mean_trace_len + 1296.0 + 1080.0


def bytes_per_sim(mean_trace_len, k_eval):
    """SURVEY.md 8(d) algorithmic bytes per simulation (packed game 64 B, packed observation 64 B, U = 7)."""
    return 204.0 * mean_trace_len - 144.0 + 1296.0 + 1000.0 * k_eval


AGENTS = {"ValueSim": 1, "ValueSimLP": 2, "DistValueSim": 4, "Vanilla": 0}     # -> BASELINE.json configs[i]


TRAINED_CHECKPOINT = os.path.join("tetris_mcts_amd", "checkpoints", "value_net_online_r05.pt")     # scripts/gpu_r05_train.sh


def run_agent(args, name, sims, warmup, steps, steady_warmup, steady_steps, ctx, checkpoint=None):
    """One agent's windows on this rank's games: `warmup` untimed moves, `steps` timed ones (barrier + synchronize on both
    sides, max over ranks, counters summed over ranks), optionally a second window later in the same games.  Returns the fields
    of a bench line (rank 0; None elsewhere) and the model."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from tetris_mcts_amd import agents, store as st, dist as tdist
    from tetris_mcts_amd.model import Model_VV
    from tetris_mcts_amd.pyTetris import Tetris
    rank, world = ctx["rank"], ctx["world"]
    G, NS = args.games, args.split
    EV_EVERY = int(os.environ.get("TM_BENCH_EVENT_EVERY", "16"))
    env_args = ((20, 10), 1, 0, 0)
    is_dist, is_vanilla = name == "DistValueSim", name == "Vanilla"
    max_nodes = args.max_nodes
    kw = {}
    if is_dist:
        # BASELINE configs[4]: the distributional head (model/model_distributional.py Net, 50 atoms over [0, 5000)), random init
        from tetris_mcts_amd.model_distributional import Model_Dist
        model = Model_Dist(atoms=50, seed=0, backend=args.backend)
        assert not args.online, "the online leg of the distributional agent is scripts/selfplay_online.py --agent DistValueSim"
        kw["model"] = model
    elif is_vanilla:
        # BASELINE configs[0]: plain UCT, random rollouts to the end of the game inside the tree kernel (CPython MT19937 per
        # game), no network; the reference's pool of 500 000 nodes is never approached at 100 simulations a move
        model = None
        max_nodes = min(max_nodes, args.vanilla_max_nodes)
        kw["random_seed"] = 0
    else:
        model = Model_VV(backend=args.backend, seed=0)  # model_vv.Net() under torch.manual_seed(0) (random init)
        if checkpoint:                                  # ... or a trained one, in the reference's checkpoint format (model/model.py:152-160)
            model.load(checkpoint, verbose=False)
        kw["model"] = model
    game = Tetris(*env_args, seed=tdist.game_seeds(20260925, G, rank), n_games=G)     # game g of rank r = game r*G + g of the job
    if not is_vanilla:
        kw.update(dict(online=True, min_visits_to_store=10, replay_cap=16384) if args.online else dict(online=False))
    agent = getattr(agents, name)(sims=sims, env=Tetris, env_args=env_args, n_games=G, max_nodes=max_nodes,
                                  n_sub=NS, ev_every=EV_EVERY, gc_slice_cycles=args.gc_slice_cycles, gc_spec_nodes=args.gc_spec_nodes,
                                  gc_cost_units=args.gc_cost_units, gc_collectors=args.gc_collectors, **kw)
    agent.update_root(game)
    torch.cuda.synchronize()
    S = agent.store
    dev = S.device
    py_loop = agent.search_model() is False     # the launch loop runs in Python (an evaluator that is not a HIP net)
    K = S.eval_slots
    NS = agent.n_sub
    tally = dict(episodes=0, lines=0, lines_all=0, game_moves=0, ended_lines=[])
    prev_lines = np.zeros(G, np.int64)      # every game's line count so far in its episode under way
    gather = dict(ms=0.0, tuples=0, bytes=0, calls=0, checksum_ok=True)

    def one_step(timed):
        action = agent.play()      # mcts(sims) + get_action (games whose reachable tree outgrew the pool restart with an empty tree)
        game.play(action)
        agent.update_root(game)   # reads game.end: one small host sync per move, as the reference's loop has
        ended = np.atleast_1d(game.end)
        cur = np.atleast_1d(game.line_clears).astype(np.int64)       # (the same host copy game.end came from: no extra sync)
        if timed:
            # the metric's second half without survivorship bias: lines cleared by ALL games in the window, per 1000 moves
            tally["lines_all"] += int((cur - prev_lines).sum())
            tally["game_moves"] += G
        prev_lines[:] = cur
        if ended.any():
            if timed:
                tally["episodes"] += int(ended.sum())
                tally["lines"] += int(cur[ended].sum())
                tally["ended_lines"] += [int(x) for x in cur[ended]]
            prev_lines[ended] = 0
            game.reset("ended")
            agent.update_root(game)
        if args.online:
            # the one exchange step of the job: this rank's freshly harvested tuples -> every rank (RCCL all-gather)
            t0 = time.perf_counter()
            keys, stats = S.replay()
            S.t["replay_count"].zero_()
            local_sum = keys.to(torch.int64).sum() + stats.view(torch.int32).to(torch.int64).sum()
            n_local = keys.shape[0]
            ka, sa = tdist.all_gather_tuples(keys.view(torch.int32), stats)
            tot = torch.stack([local_sum, torch.tensor(n_local, device=dev, dtype=torch.int64)])
            if dist.is_initialized():
                dist.all_reduce(tot)
            torch.cuda.synchronize()
            if timed:
                gather["ms"] += 1e3 * (time.perf_counter() - t0)
                gather["tuples"] += int(ka.shape[0])
                gather["bytes"] += int(ka.shape[0]) * 64
                gather["calls"] += 1
                # the gathered multiset is the union of the ranks' harvests: same count, same word sum
                got = ka.to(torch.int64).sum() + sa.view(torch.int32).to(torch.int64).sum()
                gather["checksum_ok"] &= bool(int(tot[1].item()) == int(ka.shape[0]) and int(tot[0].item()) == int(got.item()))

    CNT = ("N_EXPAND", "N_SIMS", "TRACE_SUM", "N_EVAL", "N_GC", "GC_SLICES", "N_DROPPED", "N_POOL_RESET", "PREFIX_SUM",
           "N_EVAL_SKIP", "N_EVAL_CACHED", "GC_MARK_LAUNCHES", "GC_BLOCKS", "GC_ITERS", "GC_MARK_CYC", "GC_MARK_PARTS", "GC_MARK_SHARED", "GC_CYC_LOAD", "GC_CYC_ROUNDS", "GC_CYC_WAVES", "GC_IDLE_TURNS")

    def counters():
        return {k: S.counter(k) for k in CNT}

    def measure(n_steps):
        tally.update(episodes=0, lines=0, lines_all=0, game_moves=0, ended_lines=[])
        torch.cuda.synchronize()
        S.search_stats(NS, EV_EVERY, reset=True)
        if py_loop:
            agent.loop_stats(reset=True)
        if world > 1:
            dist.barrier()
        c0 = counters()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            one_step(True)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        c1 = counters()
        ss = S.search_stats(NS, EV_EVERY, reset=False) or {}
        if py_loop:      # the launch loop ran in Python (TreeAgent.mcts): its own sampled events
            ss = agent.loop_stats(reset=True)
        err = int((S.errors() != 0).sum().item())
        d = {k: c1[k] - c0[k] for k in c0}
        tot = torch.tensor([elapsed, d["N_EXPAND"], d["N_SIMS"], d["TRACE_SUM"], d["N_EVAL"], tally["episodes"], tally["lines"], err,
                            d["N_GC"], d["GC_SLICES"], d["N_DROPPED"], ss.get("catchup_launches", 0.0), d["N_POOL_RESET"],
                            ss.get("gc_launches", 0.0), d["PREFIX_SUM"], d["N_EVAL_SKIP"], d["N_EVAL_CACHED"],
                            tally["lines_all"], tally["game_moves"], float(sum(tally["ended_lines"]) + prev_lines.sum()),
                            float(len(tally["ended_lines"]) + G), ss.get("catchup_waves", 0.0),
                            d["GC_MARK_LAUNCHES"], d["GC_BLOCKS"], d["GC_ITERS"], d["GC_MARK_CYC"], d["GC_MARK_PARTS"],
                            d["GC_MARK_SHARED"], d["GC_CYC_LOAD"], d["GC_CYC_ROUNDS"], d["GC_CYC_WAVES"], d["GC_IDLE_TURNS"]],
                           dtype=torch.float64, device=dev)
        if world > 1:
            tmax = tot[:1].clone()
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            tot[0] = tmax[0]
        keys = ("elapsed", "n_exp", "n_sims", "tr_sum", "n_eval", "episodes", "lines", "err", "n_gc", "gc_slices", "dropped",
                "catchup", "pool_resets", "gc_launches", "prefix_sum", "n_skip", "n_cached", "lines_all", "game_moves",
                "lines_under_way_sum", "episodes_under_way", "catchup_waves", "mark_launches", "mark_blocks", "mark_iters", "mark_cyc64", "mark_parts", "mark_shared", "mark_cyc_load", "mark_cyc_rounds", "mark_cyc_waves", "mark_idle_turns")
        r = dict(zip(keys, [float(x) for x in tot.cpu()]))
        r["ss"], r["steps"] = ss, n_steps
        return r

    def kernel_figures(r):
        """Per-launch durations of the two halves of a simulation step.  HIP events around the whole launch loop of every move
        give the time per simulation (evaluator + tree kernel + launch gaps); events around every EV_EVERY-th simulation give
        the two intervals themselves (`*_event_ms`, each carrying the cost of its own event records, so their plain sum
        slightly exceeds the loop's time per simulation); the figures the roofline uses are the measured intervals scaled by
        one common factor so that they add up to the loop's time per simulation."""
        ss = r["ss"]
        if py_loop and ss.get("timed"):      # Python-driven loop: the sampled intervals as they are
            nn_ev, tree_ev = ss["nn_ms_sum"] / ss["timed"], ss["tree_ms_sum"] / ss["timed"]
            return dict(nn_ms=nn_ev, tree_ms=tree_ev, nn_event_ms=nn_ev, tree_event_ms=tree_ev, per_sim_ms=nn_ev + tree_ev,
                        timed=int(ss["timed"]))
        if not ss.get("timed") or not ss.get("loop_sims"):
            return None
        nn_ev, tree_ev = ss["nn_ms_sum"] / ss["timed"], ss["tree_ms_sum"] / ss["timed"]
        per_sim = ss["loop_ms_sum"] / ss["loop_sims"]
        scale = per_sim / (nn_ev + tree_ev)
        return dict(nn_ms=nn_ev * scale, tree_ms=tree_ev * scale, nn_event_ms=nn_ev, tree_event_ms=tree_ev, per_sim_ms=per_sim,
                    timed=int(ss["timed"]))

    for _ in range(warmup):
        one_step(False)
    head = measure(steps)
    max_trace = int(S.t["gs"][:, st.GS["MAX_TRACE"]].max().item())
    phase_kcycles = {k: float(S.t["gs"][:, st.GS[k]].float().mean().item()) / 1e3
                     for k in ("CYC_BACK", "CYC_SELECT", "CYC_VERIFY", "CYC_EXPAND", "FIRST_MISS", "TRACE_LEN")}
    steady = None
    moves_done = warmup + steps
    if steady_steps > 0 and not args.online and (world == 1 or args.steady_multi):
        # the regime self-play lives in: trees fill their pools, games collect garbage every few moves (same games, later)
        for _ in range(max(0, steady_warmup - moves_done)):
            one_step(False)
        first_move = max(steady_warmup, moves_done) + 1
        steady = measure(steady_steps)
        steady["first_move"] = first_move
    store_gib = S.nbytes() / 2**30
    walk_miss = S.counter("N_WALK_MISS") / max(S.counter("TRACE_SUM"), 1)
    collective = bool(dist.is_initialized())
    backend_name = dist.get_backend() if dist.is_initialized() else None
    del agent, game, S
    torch.cuda.empty_cache()
    if rank != 0:
        return None, model

    elapsed, n_exp, n_sims, n_eval = head["elapsed"], head["n_exp"], head["n_sims"], head["n_eval"]
    mean_len = head["tr_sum"] / max(n_sims, 1.0)
    workload_key = "%s G=%d sims=%d pool=%d warmup=%d steps=%d split=%d" % (name, G, sims, max_nodes, warmup, steps, NS)
    if checkpoint:       # (other weights, other trees: the PMC passes of the random-init workload do not describe this one)
        workload_key += " checkpoint=%s" % os.path.basename(checkpoint)

    def gc_block(r):
        return {"collections": int(r["n_gc"]), "collector_launches_x_games": int(r["gc_slices"]),
                "launches_per_collection": (r["gc_slices"] / r["n_gc"]) if r["n_gc"] else None,
                "catchup_launches": int(r["catchup"]), "catchup_launches_per_move": r["catchup"] / r["steps"] / world,
                # a catch-up launch runs over the games that owe only (tm_store::game_list): their simulation waves, in units of a
                # full launch's (the evaluator draws from the dense request list and shrinks with them)
                "catchup_full_launch_equivalents_per_move": r["catchup_waves"] / float(G) / r["steps"] / world,
                "collector_only_launches": int(r["gc_launches"]), "dropped_tuples": int(r["dropped"]),
                "trees_restarted_pool_outgrown": int(r["pool_resets"]),
                # the marker (tree.hip gc_sweep_mark): launches in which a workgroup marked, per collection; blocks of child rows it
                # loaded; rounds of block-local marking per block; shader kilocycles per marking launch
                "marker": {"launches_per_collection": (r["mark_launches"] / r["n_gc"]) if r["n_gc"] else None,
                           "blocks_per_collection": (r["mark_blocks"] / r["n_gc"]) if r["n_gc"] else None,
                           "rounds_per_block": (r["mark_iters"] / r["mark_blocks"]) if r["mark_blocks"] else None,
                           "kcycles_per_block": (r["mark_cyc64"] * 0.064 / r["mark_blocks"]) if r["mark_blocks"] else None,
                           "kcycles_per_launch": (r["mark_cyc64"] * 0.064 / r["mark_launches"]) if r["mark_launches"] else None,
                           # per marking launch of a game: the workgroups it had (each owns a share of its index range), and
                           # the games its workgroup had (more games marking than marking workgroups: the time is shared)
                           "workgroups_per_launch": (r["mark_parts"] / r["mark_launches"]) if r["mark_launches"] else None,
                           "games_per_workgroup": (r["mark_shared"] / r["mark_launches"]) if r["mark_launches"] else None,
                           # a wave's kilocycles per block: waiting for the rows, marking; and per collection all in all
                           "wave_kcycles_per_block_rows": (r["mark_cyc_load"] * 0.064 / r["mark_blocks"]) if r["mark_blocks"] else None,
                           "wave_kcycles_per_block_marking": (r["mark_cyc_rounds"] * 0.064 / r["mark_blocks"]) if r["mark_blocks"] else None,
                           "wave_kcycles_per_collection": (r["mark_cyc_waves"] * 0.064 / r["n_gc"]) if r["n_gc"] else None,
                           "idle_turns_per_block": (r["mark_idle_turns"] / r["mark_blocks"]) if r["mark_blocks"] else None}}

    def request_block(r):
        """what the leaf evaluator was asked for: requests posted to the net, and those the search did without because the
        backup would not have used the output (identical results: bench.py docstring / DESIGN section 3.3)"""
        unique = r["n_eval"] + r["n_skip"] + r["n_cached"]
        return {"posted_to_the_evaluator": int(r["n_eval"]), "per_expansion": r["n_eval"] / max(r["n_exp"], 1.0),
                "leaf_parallel_children_already_visited_not_posted": int(r["n_skip"]),
                "single_leaf_observation_evaluated_before_answered_from_cache": int(r["n_cached"]),
                "fraction_not_posted": (r["n_skip"] + r["n_cached"]) / unique if unique else 0.0}

    what = ("plain UCT with random rollouts inside the tree kernel (no network)" if is_vanilla else
            "UCT with the distributional head as leaf evaluator" if is_dist else "UCT with value-net leaf evaluation")
    out = {
        "metric": "mcts_node_expansions_per_sec",
        "value": n_exp / elapsed,
        "unit": "node-expansions/s",
        "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": 1e3 * elapsed / steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": "%d games/GPU x %d sims/move, %s %s (BASELINE configs[%d]); Tetris 20x10 app=1 guideline scoring 7-bag; "
                        "node pool %d/game%s" % (G, sims, name, what, AGENTS[name], max_nodes,
                                                 "" if is_vanilla else ("; network = checkpoint %s" % os.path.basename(checkpoint)) if checkpoint
                                                 else "; network random init under manual_seed(0)"),
            "workload_key": workload_key,
            "games_per_gpu": G, "sims_per_move": sims, "agent": name, "max_nodes": max_nodes,
            "valuenet_backend": None if is_vanilla else args.backend, "sub_batches": NS, "online": bool(args.online),
            "gc_slice_cycles": args.gc_slice_cycles, "gc_spec_nodes": args.gc_spec_nodes, "gc_cost_units": args.gc_cost_units or 13, "gc_collectors": args.gc_collectors or 128,
            "checkpoint": checkpoint and os.path.relpath(checkpoint, ROOT),
        },
        "sims_per_sec": n_sims / elapsed,
        "child_steps_per_sec": 7.0 * n_exp / elapsed,
        "evaluated_states_per_sec": n_eval / elapsed,
        "requests": request_block(head),
        "mean_trace_len": mean_len,
        "max_trace_len": max_trace,
        "walk_levels_taken_over_from_the_previous_walk": head["prefix_sum"] / max(head["tr_sum"], 1.0),
        "episodes_finished": int(head["episodes"]),
        "lines_cleared_per_episode": (head["lines"] / head["episodes"]) if head["episodes"] else None,
        "lines_per_1000_moves": 1000.0 * head["lines_all"] / max(head["game_moves"], 1.0),
        "mean_lines_all_episodes_under_way": head["lines_under_way_sum"] / max(head["episodes_under_way"], 1.0),
        "lines_note": None if head["episodes"] else "no episode ends inside the timed window (random-init network, moves %d-%d "
                      "of every game); the learning curve (lines cleared per episode vs training round) is "
                      "scripts/selfplay_online.py -> profiles/*_online_learning.jsonl" % (warmup + 1, warmup + steps),
        "error_games": int(head["err"]),
        "walk_mispredicted_levels": walk_miss,
        "gc": gc_block(head),
        "store_gib_per_gpu": store_gib,
        "last_sim_phase_kcycles": phase_kcycles,
    }
    if steady is not None:
        kf = kernel_figures(steady)
        out["steady_state"] = {
            "what": "the same games later on: moves %d-%d, node pools full, every game collects garbage every few moves "
                    "(reported beside the headline window, which SURVEY.md 8(d) defines as moves %d-%d)"
                    % (steady["first_move"], steady["first_move"] + steady["steps"] - 1, warmup + 1, warmup + steps),
            "value": steady["n_exp"] / steady["elapsed"], "unit": "node-expansions/s",
            "ms_per_step": 1e3 * steady["elapsed"] / steady["steps"], "steps": steady["steps"],
            "sims_per_sec": steady["n_sims"] / steady["elapsed"],
            "mean_trace_len": steady["tr_sum"] / max(steady["n_sims"], 1.0),
            "episodes_finished": int(steady["episodes"]),
            "lines_cleared_per_episode": (steady["lines"] / steady["episodes"]) if steady["episodes"] else None,
            "lines_per_1000_moves": 1000.0 * steady["lines_all"] / max(steady["game_moves"], 1.0),
            "mean_lines_all_episodes_under_way": steady["lines_under_way_sum"] / max(steady["episodes_under_way"], 1.0),
            "requests": request_block(steady),
            "error_games": int(steady["err"]), "gc": gc_block(steady),
            "tree_kernel_ms": kf["tree_ms"] if kf else None, "value_net_ms": kf["nn_ms"] if kf else None,
        }
    if args.online:
        out["exchange"] = {"what": "all-gather of the (packed observation, value, variance, visit) tuples harvested at GC, "
                                   "once per move (tetris_mcts_amd/dist.py all_gather_tuples; `backend`: nccl = RCCL over xGMI, or gloo with --share-gpu)",
                           "backend": backend_name, "collective_ran": collective,
                           "calls": gather["calls"], "tuples": gather["tuples"], "bytes": gather["bytes"],
                           "ms_total": gather["ms"], "multiset_check": gather["checksum_ok"]}
    kf = kernel_figures(head)
    if kf:
        nn_ms, tree_ms = kf["nn_ms"], kf["tree_ms"]
        Gs = G / NS
        evals_per_launch = n_eval / max(n_sims, 1.0) * Gs
        flops = (FLOP_PER_STATE_DIST if is_dist else FLOP_PER_STATE) * evals_per_launch
        a_tf = flops / (nn_ms * 1e-3) / 1e12 if nn_ms > 0 else 0.0
        # k_eval = the states rendered for the evaluator per expansion (what was really posted)
        bps = bytes_per_sim_dist(mean_len) if is_dist else bytes_per_sim(mean_len, n_eval / max(n_exp, 1.0))
        a_gbs = bps * Gs / (tree_ms * 1e-3) / 1e9
        note = ("bytes/launch from %s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command line); null when "
                "no file was recorded for this workload" % PMC_FILE)
        tnote = ("HIP events on the launch stream: around the whole launch loop of every move (%.4f ms per simulation = evaluator "
                 "+ tree kernel + launch gaps) and around every %d-th simulation (measured intervals %.4f / %.4f ms = "
                 "avg_launch_event_ms, each carrying its own event records); avg_launch_ms = the measured intervals scaled by one "
                 "factor to add up to the loop's time per simulation, so avg_launch_ms x %d simulations <= ms_per_step.  The "
                 "rocprofv3 kernel trace of the same command is profiles/r06_kernel_stats_*.csv"
                 % (kf["per_sim_ms"], EV_EVERY, kf["nn_event_ms"], kf["tree_event_ms"], sims))
        nn_roof = {"kernel": (("distributional head (k_dn_conv + k_dn_fc: render, two convolutions, two linear layers and the softmax on the fp32 matrix cores), per launch of %d leaf slots"
                               if args.backend == "hip" else "distributional head (model_distributional.Net on PyTorch-ROCm: MIOpen / rocBLAS kernels + the request render), per evaluation of %d leaves")
                              if is_dist else "value net forward (k_vn_conv + k_vn_fc1 with the output layer folded in), per launch over %d request slots") % int(Gs * K),
                   "bound": "mfma", "achieved": a_tf, "peak": PEAK_F32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                   "frac": a_tf / PEAK_F32_MATRIX_TFLOPS,
                   "states_per_launch": evals_per_launch,
                   "traffic": pmc_traffic(["tmcts_dn::k_dn_conv", "tmcts_dn::k_dn_fc"] if is_dist else ["tmcts_vn::k_vn_conv", "tmcts_vn::k_vn_fc1"],
                                          workload_key, 2.0) if args.backend == "hip" else None,
                   "traffic_note": note + "; FETCH_SIZE x2 (wide streams)",
                   "avg_launch_ms": nn_ms, "avg_launch_event_ms": kf["nn_event_ms"], "launches_timed": kf["timed"],
                   "events_every": EV_EVERY, "timing_note": tnote}
        tree_roof = {"kernel": "k_sim_step (backup+select+expand%s), per launch of %d games" % (" + rollout" if is_vanilla else "", int(Gs)), "bound": "hbm",
                     "achieved": a_gbs, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": a_gbs / PEAK_HBM_GBPS,
                     "traffic": pmc_traffic(["tmcts::k_sim_step<%s>" % ("true" if is_vanilla else "false")], workload_key),
                     "traffic_note": note + " (narrow scattered accesses: no gfx950 correction)",
                     "avg_launch_ms": tree_ms, "avg_launch_event_ms": kf["tree_event_ms"], "algorithmic_bytes_per_sim": bps,
                     "launches_timed": kf["timed"], "events_every": EV_EVERY, "timing_note": tnote}
        if is_vanilla:
            out["roofline"] = tree_roof
        else:
            out["roofline"] = nn_roof if nn_ms >= tree_ms else tree_roof
            out["roofline_other"] = tree_roof if nn_ms >= tree_ms else nn_roof
    return out, model


def compact(line):
    """a secondary config's line inside the headline's: the figures, not the prose"""
    keep = ("value", "unit", "steps", "warmup", "ms_per_step", "sims_per_sec", "evaluated_states_per_sec", "requests", "mean_trace_len",
            "max_trace_len", "episodes_finished", "lines_cleared_per_episode", "lines_per_1000_moves", "mean_lines_all_episodes_under_way",
            "error_games", "store_gib_per_gpu", "cpu_baseline", "steady_state", "gc")
    out = {k: line[k] for k in keep if k in line}
    out["workload"] = line["config"]["workload"]
    for rk in ("roofline", "roofline_other"):
        if rk in line:
            out[rk] = {k: line[rk][k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms",
                                                "avg_launch_event_ms", "states_per_launch", "algorithmic_bytes_per_sim") if k in line[rk]}
    return out


def main():
    global np
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)     # 5 warm-up moves, then 20 timed moves
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--games", type=int, default=4096, help="games per GPU")
    ap.add_argument("--sims", type=int, default=None, help="simulations per move (default: 500; Vanilla 100, BASELINE configs[0])")
    ap.add_argument("--agent", default="ValueSim", choices=sorted(AGENTS))
    ap.add_argument("--max-nodes", type=int, default=100000)
    ap.add_argument("--vanilla-max-nodes", type=int, default=20000, help="node pool per game of the Vanilla lines")
    ap.add_argument("--backend", default="hip", choices=["hip", "torch"])
    ap.add_argument("--split", type=int, default=int(os.environ.get("TM_BENCH_SPLIT", "1")),
                    help="sub-batches of the rank's games on separate HIP streams (one sub-batch's tree kernel runs "
                    "under another's value-net kernels)")
    ap.add_argument("--gc-slice-cycles", type=int, default=150000)
    ap.add_argument("--gc-spec-nodes", type=int, default=None,
                    help="free nodes below which a game's tree is marked while it goes on simulating (default: the store's)")
    ap.add_argument("--gc-cost-units", type=int, default=0, help="bounded collection steps a launch takes on, in cost units (0: the store's default)")
    ap.add_argument("--gc-collectors", type=int, default=0, help="collector workgroups per launch (0: the store's default, 128)")
    ap.add_argument("--checkpoint", default=None, help="value-net checkpoint for the measured agent (default: random init under manual_seed(0))")
    ap.add_argument("--online", action="store_true", help="harvest training tuples at GC and all-gather them every move")
    ap.add_argument("--steady-warmup", type=int, default=75, help="the steady-state window starts after this many moves")
    ap.add_argument("--steady-steps", type=int, default=20, help="moves of the second, steady-state window (0: none)")
    ap.add_argument("--steady-multi", action="store_true", help="measure the steady-state window in multi-GPU runs too")
    ap.add_argument("--others", default="auto", choices=["auto", "none", "all"],
                    help="short lines of the other BASELINE configs (ValueSimLP, DistValueSim at 1000 sims, Vanilla at 100 sims) "
                    "inside the headline's; auto = when this is the default single-GPU ValueSim run")
    ap.add_argument("--share-gpu", action="store_true",
                    help="all ranks on cuda:0 with the gloo backend (payloads staged through host memory): exercises the multi-rank "
                    "code path on a one-GPU box; size --games so that all ranks' stores fit")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--cpu-procs", type=int, default=0, help="processes of the N-process CPU baseline (0: all host cores, at most 128)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    launched = "RANK" in os.environ      # torch.distributed.run: a process group (RCCL), even for one rank
    n_dev = torch.cuda.device_count()
    if args.gpus > 1 and not args.share_gpu and n_dev < args.gpus:
        sys.exit("bench.py --gpus %d needs %d visible GPUs, this box has %d (one process per GPU; --share-gpu puts the ranks "
                 "on one device with the gloo backend: a functional check of the multi-rank path, not a measurement)"
                 % (args.gpus, args.gpus, n_dev))
    if not launched and args.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher of N ranks of this same command line (the driver's N > 1 form,
        # `python -m torch.distributed.run ... bench.py --gpus N`, comes in with RANK set and skips this)
        import socket
        import subprocess
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py --gpus %d was launched as %d rank(s): --nproc-per-node must equal --gpus" % (args.gpus, world))
    torch.cuda.set_device(local)
    if world > 1 or launched:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if args.share_gpu:      # RCCL refuses two ranks on one device
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    import __graft_entry__ as ge
    if rank == 0 and not os.path.exists(ge.LIB):
        ge.build()
    if world > 1:
        dist.barrier()
    ctx = dict(rank=rank, world=world)
    sims = args.sims if args.sims is not None else (100 if args.agent == "Vanilla" else 500)
    args.sims = sims
    ckpt = None
    if args.checkpoint:      # (relative to the working directory, else to the repository; a missing file is an error, not a random net)
        ckpt = next((c for c in (os.path.abspath(args.checkpoint), os.path.join(ROOT, args.checkpoint)) if os.path.isfile(c)), None)
        if ckpt is None:
            sys.exit("bench.py: checkpoint %s not found" % args.checkpoint)
    out, model = run_agent(args, args.agent, sims, args.warmup, args.steps, args.steady_warmup, args.steady_steps, ctx, checkpoint=ckpt)
    others = {}
    want_others = args.others == "all" or (args.others == "auto" and world == 1 and args.agent == "ValueSim" and not args.online
                                           and args.backend == "hip" and args.split == 1 and not args.checkpoint)
    if want_others:
        # the other configurations of BASELINE.json, short windows, so that the driver's one command line shows them all
        for name, osims, w, k in (("ValueSimLP", 500, 5, 20), ("DistValueSim", 1000, 2, 5), ("Vanilla", 100, 5, 20)):
            try:
                line, omodel = run_agent(args, name, osims, w, k, 0, 0, ctx)
                if rank == 0:
                    if not args.no_cpu_baseline and name == "Vanilla":
                        line["cpu_baseline"] = cpu_baseline_vanilla(args, osims, w, k)
                    others[name] = compact(line)
                del omodel
            except Exception as e:          # a secondary line never costs the headline
                import traceback
                others[name] = {"error": repr(e), "trace": traceback.format_exc()[-400:]}
    trained = None
    ck = os.path.join(ROOT, TRAINED_CHECKPOINT)
    if want_others and os.path.isfile(ck):
        # the metric's second half: the headline's own windows (moves 6-25 and 76-95 of 4096 games x 500 simulations) under a value
        # net that this engine's online self-play trained (scripts/gpu_r06.sh online; profiles/r0[56]_online_learning*.jsonl is its
        # learning curve) - throughput under narrower, deeper-valued trees, and the lines the search clears with it
        try:
            line, tmodel = run_agent(args, "ValueSim", sims, args.warmup, args.steps, args.steady_warmup, args.steady_steps, ctx, checkpoint=ck)
            if rank == 0:
                import hashlib
                trained = compact(line)
                trained["checkpoint"] = {"file": TRAINED_CHECKPOINT, "sha256_16": hashlib.sha256(open(ck, "rb").read()).hexdigest()[:16],
                                         "format": "torch.save({'model_state_dict', 'optimizer_state_dict'}) as model/model.py:152-160 of the reference",
                                         "trained_by": "scripts/selfplay_online.py (ValueSimLP, 512 games x 200 sims, online TD fits every 50 moves), one MI355X"}
            del tmodel
        except Exception as e:
            import traceback
            trained = {"error": repr(e), "trace": traceback.format_exc()[-400:]}
    if rank == 0:
        if trained is not None:
            out["trained_net"] = trained
        # the other lines' headline numbers as scalars of the line itself (a reader that keeps only top-level scalars keeps them)
        if "steady_state" in out:
            out["steady_value"], out["steady_ms_per_step"] = out["steady_state"]["value"], out["steady_state"]["ms_per_step"]
            sg = out["steady_state"].get("gc") or {}
            out["steady_waiting_launches_per_collection"] = sg.get("launches_per_collection")
            out["steady_catchup_launches_per_move"] = sg.get("catchup_launches_per_move")
        for key, name in (("lp", "ValueSimLP"), ("dist", "DistValueSim"), ("vanilla", "Vanilla")):
            if "value" in others.get(name, {}):
                out[key + "_value"], out[key + "_ms_per_step"] = others[name]["value"], others[name]["ms_per_step"]
        if trained and "value" in trained:
            out["trained_value"], out["trained_ms_per_step"] = trained["value"], trained["ms_per_step"]
            out["trained_mean_trace_len"] = trained["mean_trace_len"]
            out["trained_lines_per_1000_moves"] = trained["lines_per_1000_moves"]
            if "steady_state" in trained:
                out["trained_steady_value"], out["trained_steady_ms_per_step"] = trained["steady_state"]["value"], trained["steady_state"]["ms_per_step"]
                tg = trained["steady_state"].get("gc") or {}
                out["trained_steady_waiting_launches_per_collection"] = tg.get("launches_per_collection")
                out["trained_steady_catchup_launches_per_move"] = tg.get("catchup_launches_per_move")
                out["trained_steady_lines_per_1000_moves"] = trained["steady_state"]["lines_per_1000_moves"]
                out["trained_steady_mean_lines_all_episodes_under_way"] = trained["steady_state"]["mean_lines_all_episodes_under_way"]
        if others:
            out["other_configs"] = others
        if world == 1 and not args.no_cpu_baseline:
            if args.agent == "DistValueSim":
                out["cpu_baseline"] = {"value": None, "unit": "node-expansions/s", "cores": 0, "kind": "port",
                                       "sample": "none: the reference has no distributional agent that runs (agents/DistValueSimOnline.py does "
                                                 "not import); the oracle's restatement is a checker, timed nowhere"}
            elif args.agent == "Vanilla":
                out["cpu_baseline"] = cpu_baseline_vanilla(args, sims, args.warmup, args.steps)
            else:
                out["cpu_baseline"] = cpu_baseline(args, model)
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline (reported only): the reference's own compiled agent on this box's host cores
# ---------------------------------------------------------------------------------------------------------------------
def _cpu_worker(kind, sims, max_nodes, lp, seconds, seed, state_dict, q, warm=5, timed=20, stride=1):
    """One process = one game at a time on one core, as the reference deploys them (cycle.sh:69-71).  Like the GPU line it
    times moves warm+1 .. warm+timed of fresh games (seeds seed, seed + stride, ...: trees of the same age, no garbage
    collection, the reference never reaches its MAX_NODES EXCEEDED state), game after game until `seconds` are spent."""
    try:
        import numpy as np
        import torch
        torch.set_num_threads(1)
        from oracle import binding as B
        from tetris_mcts_amd.model import Net
        net = Net().eval()
        net.load_state_dict(state_dict)
        acc = dict(expansions=0, sims=0, evals=0, moves=0, seconds=0.0, games=0)
        t_start = time.perf_counter()

        def torch_eval(obs):
            x = torch.from_numpy(np.asarray(obs).astype(np.float32))
            with torch.no_grad():
                return net(x.reshape(-1, 1, 20, 10))

        def play_window(step, count):
            """step() plays one move; count() returns the running (expansions, evaluated states)"""
            for _ in range(warm):
                step()
            e0, v0 = count()
            t0 = time.perf_counter()
            for _ in range(timed):
                step()
            acc["seconds"] += time.perf_counter() - t0
            e1, v1 = count()
            acc["expansions"] += e1 - e0; acc["evals"] += v1 - v0; acc["sims"] += timed * sims; acc["moves"] += timed; acc["games"] += 1

        k = 0
        while time.perf_counter() - t_start < seconds:
            gseed = seed + k * stride
            k += 1
            if kind == "reference":
                agent_mod = B.load_ref_native("agent")
                if agent_mod is None:
                    q.put(None)
                    return
                OTetris = B.oracle_pytetris().Tetris       # by path: the product has a module named pyTetris as well
                import ctypes
                ctypes.CDLL("libc.so.6").srand(1)
                g = OTetris((20, 10), 1, 0, 0, gseed)
                cnt = [0, 0]       # expand() executions = evaluator calls (MCTSAgent evaluates exactly the leaves it expands); states

                def ev(obs, cnt=cnt):
                    y = torch_eval(obs)
                    cnt[0] += 1; cnt[1] += y.shape[0]
                    return [y[:, 0].tolist(), y[:, 1].tolist()] if lp else [float(y[0, 0]), float(y[0, 1])]
                ag = agent_mod.MCTSAgent(sims, max_nodes, True, 0.999, False, ev, 0, lp)
                ag.update_root(g)

                def step(g=g, ag=ag):
                    g.play(int(ag.play()))
                    ag.update_root(g)
                    if g.end:
                        g.reset()
                        ag.update_root(g)
                play_window(step, lambda cnt=cnt: (cnt[0], cnt[1]))
            elif kind == "reference_py":
                # the reference's own Python ValueSim / ValueSimLP (agents/ValueSim.py) - needs the reference sources
                from oracle import ref_shims
                ref_shims.install()
                ref_shims.srand(1)
                cnt = [0, 0]

                def ev(states, cnt=cnt):
                    y = torch_eval(states)
                    cnt[0] += 1; cnt[1] += y.shape[0]
                    return y[:, 0].numpy(), y[:, 1].numpy()
                ag = ref_shims.make_agent("ValueSimLP" if lp else "ValueSim", sims, evaluator=ev)
                g = B.oracle_pytetris().Tetris((20, 10), 1, 0, 0, gseed)
                ag.update_root(g)

                def step(g=g, ag=ag):
                    g.play(int(ag.play()))
                    ag.update_root(g)
                    if g.end:
                        g.reset()
                        ag.update_root(g)
                play_window(step, lambda cnt=cnt: (cnt[0], cnt[1]))
            else:
                from tetris_mcts_amd.model import PARAM_ORDER      # (a state_dict lists out_ubound / out_lbound first)
                params = torch.cat([state_dict[kk].reshape(-1).float() for kk in PARAM_ORDER]).numpy()
                g = B.Game(seed=gseed)
                a = B.Agent(1 if lp else 0, max_nodes=max_nodes, evaluator="valuenet", params=params)
                a.update_root(g)

                def step(g=g, a=a):
                    g.play(a.play(sims))
                    a.update_root(g)
                    if g.end:
                        g.reset()
                        a.update_root(g)
                play_window(step, lambda a=a: (a.n_expand, a.n_expand))
        q.put(acc)
    except Exception as e:  # reported, never fatal for the benchmark line
        import traceback
        q.put(dict(error=repr(e) + " " + traceback.format_exc()[-300:]))


def _cpu_run(kind, nproc, args, state_dict, seconds):
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    lp = args.agent != "ValueSim"
    procs = [ctx.Process(target=_cpu_worker, args=(kind, args.sims, args.max_nodes, lp, seconds, 20260925 + i, state_dict, q,
                                                   args.warmup, args.steps, nproc))
             for i in range(nproc)]
    t0 = time.perf_counter()
    for p in procs:
        p.start()
    # every worker reports once; one that died (or hangs) must not hold the benchmark line up: a fixed deadline
    res, deadline = [], time.perf_counter() + 3.0 * seconds + 180.0      # (a game's window is finished once it is begun)
    while len(res) < len(procs) and time.perf_counter() < deadline:
        try:
            res.append(q.get(timeout=2.0))
        except Exception:
            if not any(p.is_alive() for p in procs):
                break
    for p in procs:
        p.join(timeout=5)
        if p.is_alive():
            p.kill()
    wall = time.perf_counter() - t0
    ok = [r for r in res if r and "error" not in r]
    errs = [r["error"] for r in res if r and "error" in r]
    if not ok:
        return None, errs
    ok = [r for r in ok if r["seconds"] > 0]
    if not ok:
        return None, errs
    rate = sum(r["expansions"] / r["seconds"] for r in ok)
    return dict(value=rate, procs_ok=len(ok), expansions=sum(r["expansions"] for r in ok), sims=sum(r["sims"] for r in ok),
                sims_per_sec=sum(r["sims"] / r["seconds"] for r in ok), moves=sum(r["moves"] for r in ok),
                games=sum(r["games"] for r in ok), seconds_each=seconds, wall=wall), errs


def cpu_baseline(args, model):
    """the reference timed on this box's host cores, reported only.  kind "reference" = the reference's
    own all-C++ MCTSAgent (agents/cppmodule/agent.cpp compiled in place into oracle/_ref/ - the ValueSimC / best-case
    path) with a single-thread torch CPU `Net` as its evaluator, (a) 1 process on 1 core and (b) one process per host
    core (independent games, the reference's own deployment, cycle.sh:69-71).  `value` = expand() executions per second
    (= evaluator calls of MCTSAgent: it evaluates exactly the leaves it expands).  The reference's *Python* ValueSim
    needs the reference sources at run time (/root/reference is not on the GPU box), so it is timed only where the
    sources are (DESIGN.md section 5 lists that number).  Fallback kind "port": the oracle's C restatement."""
    sd = {k: v.detach().cpu() for k, v in model.model.state_dict().items()}
    ncpu = os.cpu_count() or 1
    quota = None        # a container may own fewer cores than it sees (cgroup v2 cpu.max / v1 cfs quota)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()
            quota = None if q == "max" else float(q) / float(per)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            quota = q / per if q > 0 else None
        except Exception:
            quota = None
    try:
        ncpu_aff = len(os.sched_getaffinity(0))
    except Exception:
        ncpu_aff = ncpu
    usable = int(min(ncpu_aff, quota) if quota else ncpu_aff)
    nproc = args.cpu_procs or max(1, min(usable - 1, 128))     # one core stays with this process
    sample = ("per process: fresh games (seed 20260925 + i, + n_procs, ...), %d sims/move, pool %d, the GPU line's own window - moves "
              "%d-%d of every game timed, the first %d untimed - game after game for %.0f s; same network weights")
    kind = "reference"
    one, errs = _cpu_run(kind, 1, args, sd, args.cpu_seconds)
    if one is None:
        kind = "port"
        one, errs = _cpu_run(kind, 1, args, sd, args.cpu_seconds)
    if one is None:
        return {"value": None, "unit": "node-expansions/s", "cores": 0, "kind": "port", "sample": "unavailable: %s" % errs[:1]}
    many, errs2 = _cpu_run(kind, nproc, args, sd, args.cpu_seconds) if nproc > 1 else (None, [])
    out = {"value": (many or one)["value"], "unit": "node-expansions/s", "cores": (many["procs_ok"] if many else 1), "kind": kind,
           "sample": (sample % (args.sims, args.max_nodes, args.warmup + 1, args.warmup + args.steps, args.warmup, args.cpu_seconds)) +
                     ("; the reference's compiled MCTSAgent (agent.cpp, LP=%s) + torch CPU Net, 1 thread per process" % (args.agent != "ValueSim")
                      if kind == "reference" else "; oracle C restatement incl. its fp32 value net"),
           "host_cpus": ncpu, "cpu_affinity": ncpu_aff, "cgroup_cpu_quota": quota,
           "one_core": {"value": one["value"], "sims_per_sec": one["sims_per_sec"], "moves": one["moves"], "games": one["games"]},
           "all_cores": None if many is None else {"value": many["value"], "procs": many["procs_ok"], "sims_per_sec": many["sims_per_sec"],
                                                   "per_core": many["value"] / max(many["procs_ok"], 1), "wall_s": many["wall"],
                                                   "games": many["games"]}}
    # the agent BASELINE configs[1] names, the reference's Python ValueSim: runs only where the reference sources are
    if os.path.isdir(os.environ.get("TETRIS_MCTS_REFERENCE", "/root/reference")):
        py, errs3 = _cpu_run("reference_py", 1, args, sd, args.cpu_seconds)
        out["python_agent_one_core"] = (None if py is None else
                                        {"value": py["value"], "sims_per_sec": py["sims_per_sec"], "moves": py["moves"], "games": py["games"],
                                         "what": "agents/%s.py of the reference, imported unmodified (oracle/ref_shims.py), same window" % args.agent})
        errs2 = errs2 + errs3
    else:
        out["python_agent_one_core"] = None
        out["python_agent_note"] = ("the reference's Python agent needs the reference sources, which are not on this box; timed in "
                                    "the build container: DESIGN.md section 5")
    if errs or errs2:
        out["worker_errors"] = (errs + errs2)[:3]
    return out


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline of BASELINE configs[0]: "1 game, Vanilla agent, 100 sims/move, random-rollout (no NN) on CPU reference path"
# ---------------------------------------------------------------------------------------------------------------------
def _cpu_worker_vanilla(kind, sims, max_nodes, seconds, seed, q, warm=5, timed=20, stride=1):
    """One process = one game at a time on one core; moves warm+1 .. warm+timed of fresh games, game after game.
    kind "reference_c":  the reference's compiled MCTSAgent with evaluator type 1 = a Python random playout (agents/VanillaC.py:5-14)
         "reference_py": the reference's Python Vanilla (agents/Vanilla.py:17-64), needs the reference sources
         "port":         the oracle's C restatement of Vanilla (agent_oracle.c kind 4)"""
    try:
        import random
        from oracle import binding as B
        acc = dict(expansions=0, sims=0, moves=0, seconds=0.0, games=0, rollout_steps=0)
        t_start = time.perf_counter()

        def play_window(step, count):
            for _ in range(warm):
                step()
            e0 = count()
            t0 = time.perf_counter()
            for _ in range(timed):
                step()
            acc["seconds"] += time.perf_counter() - t0
            acc["expansions"] += count() - e0; acc["sims"] += timed * sims; acc["moves"] += timed; acc["games"] += 1

        k = 0
        while time.perf_counter() - t_start < seconds:
            gseed = seed + k * stride
            k += 1
            random.seed(gseed)
            if kind == "reference_c":
                agent_mod = B.load_ref_native("agent")
                if agent_mod is None:
                    q.put(None)
                    return
                import ctypes
                ctypes.CDLL("libc.so.6").srand(1)
                g = B.oracle_pytetris().Tetris((20, 10), 1, 0, 0, gseed)
                cnt = [0]

                def random_playout(game, cnt=cnt):            # agents/VanillaC.py:5-8; one call per expanded leaf
                    cnt[0] += 1
                    while not game.end:
                        game.play(random.randint(0, 7))
                    return game.score, 1e5
                ag = agent_mod.MCTSAgent(sims, max_nodes, True, 0.99, False, random_playout, 1, False)
                ag.update_root(g)

                def step(g=g, ag=ag):
                    g.play(int(ag.play()))
                    ag.update_root(g)
                    if g.end:
                        g.reset()
                        ag.update_root(g)
                play_window(step, lambda cnt=cnt: cnt[0])
            elif kind == "reference_py":
                from oracle import ref_shims
                ref_shims.install()
                ref_shims.srand(1)
                ag = ref_shims.make_agent("Vanilla", sims, max_nodes=max_nodes)
                cnt = [0]
                inner = ag.expand

                def counted(game, cnt=cnt, inner=inner):
                    cnt[0] += 1
                    return inner(game)
                ag.expand = counted
                g = B.oracle_pytetris().Tetris((20, 10), 1, 0, 0, gseed)
                ag.update_root(g)

                def step(g=g, ag=ag):
                    g.play(int(ag.play()))
                    ag.update_root(g)
                    if g.end:
                        g.reset()
                        ag.update_root(g)
                play_window(step, lambda cnt=cnt: cnt[0])
            else:
                g = B.Game(seed=gseed)
                a = B.Agent(4, max_nodes=max_nodes, gamma=0.99, low=5)
                a.set_python_random_state(random.Random(gseed).getstate())
                a.update_root(g)

                def step(g=g, a=a):
                    g.play(a.play(sims))
                    a.update_root(g)
                    if g.end:
                        g.reset()
                        a.update_root(g)
                play_window(step, lambda a=a: a.n_expand)
        q.put(acc)
    except Exception as e:
        import traceback
        q.put(dict(error=repr(e) + " " + traceback.format_exc()[-300:]))


def cpu_baseline_vanilla(args, sims, warm, timed):
    """configs[0] on this box's host cores, reported only: the reference's compiled agent with its Python playout (VanillaC,
    `kind` "reference": oracle/_ref travels with the snapshot), the reference's Python Vanilla where its sources are, and the
    oracle's C restatement (a scalar port: an upper bound of what one core does with this algorithm)."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    max_nodes = min(args.max_nodes, args.vanilla_max_nodes)

    def run(kind, nproc, seconds):
        q = ctx.Queue()
        procs = [ctx.Process(target=_cpu_worker_vanilla, args=(kind, sims, max_nodes, seconds, 20260925 + i, q, warm, timed, nproc))
                 for i in range(nproc)]
        for p in procs:
            p.start()
        res, deadline = [], time.perf_counter() + 4.0 * seconds + 120.0
        while len(res) < len(procs) and time.perf_counter() < deadline:
            try:
                res.append(q.get(timeout=2.0))
            except Exception:
                if not any(p.is_alive() for p in procs):
                    break
        for p in procs:
            p.join(timeout=5)
            if p.is_alive():
                p.kill()
        ok = [r for r in res if r and "error" not in r and r["seconds"] > 0]
        errs = [r["error"] for r in res if r and "error" in r]
        if not ok:
            return None, errs
        return dict(value=sum(r["expansions"] / r["seconds"] for r in ok), sims_per_sec=sum(r["sims"] / r["seconds"] for r in ok),
                    procs=len(ok), moves=sum(r["moves"] for r in ok), games=sum(r["games"] for r in ok)), errs

    secs = max(3.0, args.cpu_seconds / 2)
    try:
        usable = len(os.sched_getaffinity(0))
    except Exception:
        usable = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            qv, per = f.read().split()
            if qv != "max":
                usable = int(min(usable, float(qv) / float(per)))
    except Exception:
        pass
    nproc = args.cpu_procs or max(1, min(usable - 1, 128))
    kind = "reference"
    one, errs = run("reference_c", 1, secs)
    many = None
    if one is None:
        kind = "port"
        one, errs = run("port", 1, secs)
    elif nproc > 1:
        many, e2 = run("reference_c", nproc, secs)
        errs += e2
    if one is None:
        return {"value": None, "unit": "node-expansions/s", "cores": 0, "kind": "port", "sample": "unavailable: %s" % errs[:1]}
    out = {"value": (many or one)["value"], "unit": "node-expansions/s", "cores": (many["procs"] if many else 1), "kind": kind,
           "sample": "per process: fresh games (seed 20260925 + i, + n_procs, ...), %d sims/move, pool %d, moves %d-%d of every game "
                     "timed, game after game for %.0f s; %s" % (sims, max_nodes, warm + 1, warm + timed, secs,
                     "the reference's compiled MCTSAgent with evaluator type 1, a Python random playout (agents/VanillaC.py)"
                     if kind == "reference" else "oracle C restatement of agents/Vanilla.py"),
           "one_core": one, "all_cores": many}
    if kind == "reference":
        port, e3 = run("port", 1, secs)
        out["oracle_port_one_core"] = port
    if os.path.isdir(os.environ.get("TETRIS_MCTS_REFERENCE", "/root/reference")):
        py, e4 = run("reference_py", 1, secs)
        out["python_agent_one_core"] = py and dict(py, what="agents/Vanilla.py of the reference, imported unmodified (oracle/ref_shims.py)")
        errs += e4
    else:
        out["python_agent_one_core"] = None
    if errs:
        out["worker_errors"] = errs[:3]
    return out


if __name__ == "__main__":
    main()
