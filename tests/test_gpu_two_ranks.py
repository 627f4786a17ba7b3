"""The product as TWO processes side by side (BASELINE configs[3]'s code path: one process per shard of the games, the tuple
all-gather, the data-parallel fit) - on the one GPU a test box has.  RCCL refuses two ranks on one device, so the group is gloo
and dist.py stages the payloads through host memory; everything else is what an 8-GPU job runs: per-process kernel attributes,
`weights_epoch`, the native launch loop's streams, the pools' initialisation, the request-list parity protocol, twice, in two
address spaces, interleaved on the same device.

  (a) every game of the job plays the same actions with the same root statistics (bytes) whether the job is one rank of 4096
      games or two ranks of 2048 ("per-game trajectories independent of P"), ON THE GPU ENGINE, through
      garbage collections;
  (b) the all-gathered tuples of every move are the single-rank harvest of that move (multiset; 8(e): "multiset of gathered
      tuples equals single-process multiset");
  (c) one data-parallel train_data on the union leaves identical weights on both ranks.
And bench.py itself as two ranks (`--gpus 2 --share-gpu`): the barrier / max-over-ranks / summed-counter path of the line.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G_JOB, BASE, SIMS, MOVES, POOL = 4096, 20260925, 500, 10, 6000


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _play(rank, world, gather):
    """the games of `rank` in a job of `world` ranks: per move (actions, root statistics, this move's tuples after `gather`)"""
    from tetris_mcts_amd import agents, dist as tdist
    from tetris_mcts_amd.model import Model_VV
    from tetris_mcts_amd.pyTetris import Tetris
    G = G_JOB // world
    model = Model_VV(backend="hip", seed=0)
    game = Tetris((20, 10), 1, 0, 0, seed=tdist.game_seeds(BASE, G, rank), n_games=G)
    agent = agents.ValueSim(sims=SIMS, env=Tetris, env_args=game.env_args, n_games=G, max_nodes=POOL, model=model, online=True,
                            min_visits_to_store=10, replay_cap=POOL)
    agent.update_root(game)
    S = agent.store
    acts, stats, tuples = [], [], []
    for _ in range(MOVES):
        a = agent.play()
        acts.append(np.array(a, copy=True))
        stats.append(agent.get_stats())
        game.play(a)
        agent.update_root(game)
        if np.atleast_1d(game.end).any():
            game.reset("ended")
            agent.update_root(game)
        keys, st = S.replay()
        S.t["replay_count"].zero_()
        ka, sa = gather(keys.view(torch.int32), st)
        tuples.append(np.concatenate([ka.cpu().numpy().view(np.int32), sa.cpu().numpy().view(np.int32)], axis=1))
    info = dict(collections=S.counter("N_GC"), dropped=S.counter("N_DROPPED"), resets=S.counter("N_POOL_RESET"),
                errors=int((S.errors() & ~1).ne(0).sum().item()))
    return agent, np.stack(acts), np.stack(stats), tuples, info


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)                          # both ranks on the same device
    import tempfile
    os.chdir(tempfile.mkdtemp())                      # (rank 0's train_data writes ./pytorch_model/model_checkpoint)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tetris_mcts_amd import dist as tdist
    agent, acts, stats, tuples, info = _play(rank, world, tdist.all_gather_tuples)
    # (c) one data-parallel fit on the union of everything gathered (every rank holds the same rows)
    allt = torch.from_numpy(np.concatenate(tuples)).cuda()
    data = tdist.training_arrays(allt[:, :12].contiguous(), allt[:, 12:].contiguous().view(torch.float32))
    gen = torch.Generator(device="cuda").manual_seed(99)
    res = agent.model.train_data(list(data), batch_size=256, iters_per_val=4, max_iters=8, generator=gen, log=False)
    flat = agent.model.flat_params().cpu().numpy()
    # (d) the distributional agent's exchange (dist.all_gather_rows: packed observation, distribution, visits) on DEVICE tensors,
    # staged through host memory by the gloo group; the second call with a rank that has harvested nothing
    rows = [tuple(c.cpu().numpy() for c in tdist.all_gather_rows(*_dist_rows(rank, n))) for n in (5 + 3 * rank, 4 * rank)]
    q.put((rank, acts, stats, tuples, info, flat, res["iters"], rows))
    dist.barrier()
    dist.destroy_process_group()


def _dist_rows(rank, n):
    g = torch.Generator().manual_seed(1000 + rank)
    return (torch.randint(-2 ** 31, 2 ** 31 - 1, (n, 12), generator=g, dtype=torch.int64).to(torch.int32).cuda(),
            torch.rand(n, 64, generator=g).cuda(), torch.rand(n, generator=g).cuda())


def _rows_sorted(a):
    return a[np.lexsort(a.T[::-1])] if len(a) else a


def test_two_ranks_on_one_gpu_equal_the_single_rank_job():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    # the single-rank job meanwhile, in this process: a third address space on the same device
    agent, acts1, stats1, tuples1, info1 = _play(0, 1, lambda k, s: (k, s))
    del agent
    torch.cuda.empty_cache()
    res = sorted([q.get(timeout=900) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert info1["collections"] > G_JOB // 2 and info1["dropped"] == 0 and info1["errors"] == 0, info1      # the pools did run dry
    # (a) trajectories: game g of rank r = game r * 2048 + g of the job
    acts2 = np.concatenate([r[1] for r in res], axis=1)
    stats2 = np.concatenate([r[2] for r in res], axis=1)
    assert acts2.shape == acts1.shape == (MOVES, G_JOB)
    bad = np.argwhere(acts2 != acts1)
    assert len(bad) == 0, ("first differing (move, game)", bad[:5])
    assert stats2.tobytes() == stats1.tobytes()
    assert sum(r[4]["collections"] for r in res) == info1["collections"]
    assert sum(r[4]["resets"] for r in res) == info1["resets"] and all(r[4]["dropped"] == 0 and r[4]["errors"] == 0 for r in res)
    # (b) every move's gathered tuples, on both ranks, = that move's single-rank harvest
    n_tuples = 0
    for m in range(MOVES):
        want = _rows_sorted(tuples1[m])
        n_tuples += len(want)
        for r in res:
            assert r[3][m].shape == tuples1[m].shape, (m, r[0], r[3][m].shape, tuples1[m].shape)
            assert _rows_sorted(r[3][m]).tobytes() == want.tobytes(), (m, r[0])
    assert n_tuples > 1000
    # (c) the replicas took the same steps
    assert res[0][6] == res[1][6] == 8 and res[0][5].tobytes() == res[1][5].tobytes()
    # (d) rows of three columns, rank after rank, the same on both ranks - also when rank 0 has none
    for call, sizes in enumerate(((5, 8), (0, 4))):
        want = [np.concatenate([_dist_rows(r, n)[c].cpu().numpy() for r, n in enumerate(sizes)]) for c in range(3)]
        for r in res:
            for c in range(3):
                assert r[7][call][c].dtype == want[c].dtype and r[7][call][c].tobytes() == want[c].tobytes(), (call, r[0], c)
    m0 = __import__("tetris_mcts_amd.model", fromlist=["Model_VV"]).Model_VV(backend="hip", seed=0)
    assert np.abs(res[0][5] - m0.flat_params().cpu().numpy()).max() > 1e-4           # ... and they did move


def test_bench_as_two_ranks_sharing_the_gpu():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--online", "--games", "256", "--max-nodes", "6000",
           "--steps", "3", "--warmup", "9", "--no-cpu-baseline", "--steady-steps", "0"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["error_games"] == 0
    # counters are summed over the ranks: 2 x 256 games x 500 simulations x 3 moves
    assert abs(d["sims_per_sec"] * d["ms_per_step"] * 1e-3 * 3 - 2 * 256 * 500 * 3) < 1.0
    ex = d["exchange"]
    assert ex["backend"] == "gloo" and ex["collective_ran"] and ex["multiset_check"] is True and ex["calls"] == 3
