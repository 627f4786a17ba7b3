"""The one exchange step of the job on real GPUs: RCCL (`nccl` backend) all-gather of the (packed observation, value,
variance, visit) tuples that garbage collections harvest, two ranks on two GPUs, one sharded self-play each
(BASELINE configs[3]; tetris_mcts_amd/dist.py).  Skipped on a box with fewer than two GPUs - the gloo twin of this test
(tests/test_dist.py) runs everywhere."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    from gpu_helpers import hash_eval_torch
    from tetris_mcts_amd import agents, dist as tdist
    from tetris_mcts_amd.pyTetris import Tetris
    G_total = 12
    start, count = tdist.shard_range(G_total, rank, world)
    game = Tetris((20, 10), 1, 0, 0, seed=300 + start, n_games=count)
    agent = agents.ValueSim(sims=40, env=Tetris, env_args=game.env_args, n_games=count, max_nodes=3000,
                            evaluator=hash_eval_torch, online=True, min_visits_to_store=3, replay_cap=8192)
    agent.update_root(game)
    for m in range(70):
        act = agent.play()
        game.play(act)
        agent.update_root(game)
        if game.end.any():
            game.reset("ended")
            agent.update_root(game)
    keys, stats = agent.store.replay()
    ka, sa = tdist.all_gather_tuples(keys.view(torch.int32), stats)
    dropped = agent.store.counter("N_DROPPED")
    q.put((rank, keys.cpu().numpy(), stats.cpu().numpy(), ka.cpu().numpy(), sa.cpu().numpy(), dropped))
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_of_harvested_tuples_nccl_world2():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL); the gloo twin is tests/test_dist.py")
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    exp_k = np.concatenate([r[1].view(np.int32).reshape(-1, 12) for r in res])
    exp_s = np.concatenate([r[2] for r in res])
    assert len(exp_k) > 0
    for r in res:
        assert np.array_equal(r[3], exp_k) and r[4].tobytes() == exp_s.tobytes()      # the rank-ordered union, on every rank
        assert r[5] == 0                                                                 # nothing was dropped on the way
