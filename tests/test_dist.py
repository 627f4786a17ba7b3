"""Multi-process path on CPU (gloo, world_size 2): sharding covers the games exactly once and the all-gather of
variable-length training tuples returns the rank-ordered union on every rank (= the single-process multiset)."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _tuples(rank):
    rng = np.random.default_rng(100 + rank)
    n = [5, 0, 9, 3][rank % 4] + 2 * rank
    keys = rng.integers(-2**31, 2**31 - 1, size=(n, 12), dtype=np.int64).astype(np.int32)
    stats = rng.random((n, 4)).astype(np.float32)
    return torch.from_numpy(keys), torch.from_numpy(stats)


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from tetris_mcts_amd import dist as tdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    k, s = _tuples(rank)
    ka, sa = tdist.all_gather_tuples(k, s)
    # the online path of ValueSimC: episode counters are summed, every rank feeds the same union into its replay
    # memory and therefore takes the same decision and trains on the same set
    from tetris_mcts_amd.replay import ReplayMemory
    episodes = tdist.all_sum(3 + rank)
    mem = ReplayMemory(accumulation_policy=2, memory_size=6, episodes_per_train=50)
    out = mem.absorb(ka, sa, episodes)
    # the job's replay memory (ValueSim.train_nodes): 9 of the ranks' 5 + 2 tuples would fit - 6 is the memory's size
    kj, sj = tdist.job_memory(6, ka, sa)
    assert kj.shape[0] == 6 and torch.equal(kj, ka[:6]) and torch.equal(sj, sa[:6])
    q.put((rank, ka.numpy(), sa.numpy(), tdist.shard_range(4099, rank, world), episodes, out[0].numpy(), tdist.rank()))
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_tuples_gloo_world2():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    exp_k = np.concatenate([_tuples(r)[0].numpy() for r in range(world)])
    exp_s = np.concatenate([_tuples(r)[1].numpy() for r in range(world)])
    shards = []
    for rank, ka, sa, sh, episodes, train_keys, my_rank in sorted(res, key=lambda t: t[0]):
        assert np.array_equal(ka, exp_k) and sa.tobytes() == exp_s.tobytes()
        assert episodes == 7 and my_rank == rank
        assert np.array_equal(train_keys, exp_k[:6])       # policy 2: the memory (6 of the 7 tuples) filled up -> train
        assert len(exp_k) == 7                             # (the ranks' 5 + 2 tuples overflow a memory of 6: job_memory above cut them, rank 0's first)
        shards.append(sh)
    assert shards[0][0] == 0 and shards[0][0] + shards[0][1] == shards[1][0] and shards[1][0] + shards[1][1] == 4099


def test_shard_range_partitions():
    from tetris_mcts_amd.dist import shard_range
    for n, w in ((4096, 8), (32768, 8), (10, 3), (7, 8)):
        parts = [shard_range(n, r, w) for r in range(w)]
        assert parts[0][0] == 0 and sum(c for _, c in parts) == n
        for a, b in zip(parts, parts[1:]):
            assert a[0] + a[1] == b[0]


def test_render_observations_matches_oracle(oracle):
    from tetris_mcts_amd.dist import render_observations
    rng = np.random.default_rng(3)
    keys, ref = [], []
    for seed in range(40):
        g = oracle.Game(seed=seed)
        for _ in range(int(rng.integers(0, 120))):
            g.play(int(rng.integers(0, 7)))
        keys.append(g.packed_obs().view(np.int32).reshape(12))
        ref.append(g.getState())
    out = render_observations(torch.from_numpy(np.stack(keys))).numpy().reshape(-1, 20, 10)
    assert np.array_equal(out, np.stack(ref).astype(np.float32))


def _dp_worker(rank, world, port, q):
    import torch.distributed as dist
    from tetris_mcts_amd import train as T
    from tetris_mcts_amd.model import Net
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net, data = _dp_setup()
    gen = torch.Generator().manual_seed(1234)       # one sampling stream for both ranks: each takes every second index
    res = T.train_data(net, T.Yogi(net.parameters(), lr=1e-3, eps=1e-3), data, batch_size=32, iters_per_val=4, max_iters=8,
                       generator=gen, log=False)
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    q.put((rank, flat.numpy(), res["iters"], res["best_validation"]))
    dist.barrier()
    dist.destroy_process_group()


def _dp_setup():
    from tetris_mcts_amd.model import Net
    torch.manual_seed(0)
    net = Net()
    g = torch.Generator().manual_seed(5)
    n = 200
    data = [torch.randint(-1, 2, (n, 1, 20, 10), generator=g).float(), torch.rand(n, 1, generator=g) * 50,
            torch.rand(n, 1, generator=g) * 100 + 1, torch.randint(1, 30, (n, 1), generator=g).float()]
    return net, data


def test_data_parallel_training_gloo_world2():
    """train_data with two ranks: both draw the same 32 indices per iteration and each takes every second one, the
    gradients are averaged by one all-reduce, the replicas end bit-identical - and the run is the single-process run with
    batch_size 32 and the same sampling seed (same samples per iteration; the two half-batch means are averaged instead of
    one 32-sample mean, so equal up to float rounding, not bit for bit)."""
    from tetris_mcts_amd import train as T
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1].tobytes() == res[1][1].tobytes() and res[0][2:] == res[1][2:]
    net, data = _dp_setup()
    start = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).numpy().copy()
    single = T.train_data(net, T.Yogi(net.parameters(), lr=1e-3, eps=1e-3), data, batch_size=32, iters_per_val=4, max_iters=8,
                          generator=torch.Generator().manual_seed(1234), log=False)
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).numpy()
    assert single["iters"] == res[0][2]
    assert np.abs(flat - start).max() > 1e-3                       # the run moved the weights ...
    assert np.abs(flat - res[0][1]).max() < 2e-5                   # ... and both runs moved them the same way


# ---- a game's trajectory does not depend on the number of ranks -------------------------------------------------------
_JOB_GAMES, _JOB_BASE, _JOB_MOVES, _JOB_SIMS = 8, 20260925, 5, 60


def _play_shard(oracle_binding, rank, world):
    """The games of `rank` in a job of `world` ranks (dist.game_seeds: game g of rank r = game r * G + g of the job), each
    with its own oracle agent: per game the list of (action, score) over the first moves."""
    from tetris_mcts_amd import dist as tdist
    per = _JOB_GAMES // world
    out = {}
    for g, seed in enumerate(tdist.game_seeds(_JOB_BASE, per, rank)):
        game = oracle_binding.Game(seed=int(seed))
        agent = oracle_binding.Agent(0, max_nodes=4000)       # ValueSim numerics, hash evaluator
        agent.update_root(game)
        traj = []
        for _ in range(_JOB_MOVES):
            a = agent.play(_JOB_SIMS)
            game.play(a)
            agent.update_root(game)
            traj.append((int(a), int(game.score)))
        out[rank * per + g] = traj
    return out


def _traj_worker(rank, world, port, q):
    import sys
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import binding as B
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = _play_shard(B, rank, world)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    q.put((rank, gathered))
    dist.barrier()
    dist.destroy_process_group()


def test_trajectories_do_not_depend_on_world_size(oracle):
    """8 games as one process and as 2 ranks of 4 (gloo): every game - identified by its index in the job - plays the
    same actions to the same scores.  (The reference's agents run on the CPU oracle here; the GPU engine is held to the
    oracle game by game in the -m gpu tests, so the property carries over.)"""
    single = _play_shard(oracle, 0, 1)
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_traj_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, gathered in res:
        merged = {}
        for part in gathered:
            merged.update(part)
        assert sorted(merged) == list(range(_JOB_GAMES))
        assert merged == single, rank


def test_data_parallel_training_rejects_unequal_shards():
    """batch_size must split evenly over the ranks: checked before any collective is issued."""
    import pytest
    from tetris_mcts_amd import train as T
    T.check_data_parallel_batch(7, 1)
    T.check_data_parallel_batch(1024, 8)
    with pytest.raises(ValueError):
        T.check_data_parallel_batch(1025, 8)
    with pytest.raises(ValueError):
        T.check_data_parallel_batch(4, 8)       # a rank would get an empty shard


def _rows(rank):
    rng = np.random.default_rng(700 + rank)
    n = [4, 0, 7][rank % 3] + rank
    keys = torch.from_numpy(rng.integers(-2**31, 2**31 - 1, size=(n, 12), dtype=np.int64).astype(np.int32))
    d = rng.random((n, 64)).astype(np.float32)
    d[:, 50:] = 0
    return keys, torch.from_numpy(d), torch.from_numpy(rng.integers(8, 400, size=n).astype(np.float32))


def _rows_worker(rank, world, port, q):
    """the distributional agent's exchange + fit: harvested (observation, distribution, visits) rows of unequal counts per rank
    -> the same union on every rank -> the same data-parallel fit of the head (model_distributional.Model_Dist) on every rank"""
    import torch.distributed as dist
    from tetris_mcts_amd import dist as tdist
    from tetris_mcts_amd.model_distributional import Model_Dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    k, d, v = _rows(rank)
    ka, da, va = tdist.all_gather_rows(k, d, v)
    m = Model_Dist(atoms=50, device="cpu", seed=4, backend="torch")
    n = ka.shape[0]
    states = torch.zeros(n, 1, 22, 10)
    states[:, :, 2:, :] = tdist.render_observations(ka)
    tgt = da[:, :50] / da[:, :50].sum(1, keepdim=True)
    m.train_data([states, tgt, va.reshape(-1, 1)], batch_size=8, iters_per_val=5, max_iters=10, log=False, validation_fraction=0.0)
    flat = torch.cat([p.detach().reshape(-1) for p in m.model.parameters()])
    q.put((rank, ka.numpy(), da.numpy(), va.numpy(), flat.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_rows_and_distributional_fit_gloo_world2():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rows_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    exp = [np.concatenate([_rows(r)[i].numpy() for r in range(world)]) for i in range(3)]
    for r in res:
        assert np.array_equal(r[1], exp[0]) and r[2].tobytes() == exp[1].tobytes() and r[3].tobytes() == exp[2].tobytes()
    assert res[0][4].tobytes() == res[1][4].tobytes()          # the replicas took the same steps


def _empty_rank_worker(rank, world, port, q):
    import torch.distributed as dist
    from tetris_mcts_amd import dist as tdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    k, d, v = _rows(0)
    if rank == 1:         # this rank's games have not collected yet: nothing harvested
        k, d, v = k[:0], d[:0], v[:0]
    ka, da, va = tdist.all_gather_rows(k, d, v)
    tk, ts = _tuples(0)
    if rank == 1:
        tk, ts = tk[:0], ts[:0]
    tka, tsa = tdist.all_gather_tuples(tk, ts)
    # nobody has anything: both calls still return (and nobody hangs in a collective the others skipped)
    e = tdist.all_gather_rows(k[:0], d[:0], v[:0])
    q.put((rank, ka.numpy(), da.numpy(), va.numpy(), tka.numpy(), tsa.numpy(), [tuple(x.shape) for x in e]))
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_with_a_rank_that_has_no_rows_gloo_world2():
    """Collections are not synchronised over the ranks: a rank with nothing harvested yet takes part in the exchange with zero
    rows (round-4 advisor: reshape(0, -1) raised on that rank before the collectives and the other ranks hung)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_empty_rank_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    k, d, v = _rows(0)
    tk, ts = _tuples(0)
    for r in res:
        assert np.array_equal(r[1], k.numpy()) and r[2].tobytes() == d.numpy().tobytes() and r[3].tobytes() == v.numpy().tobytes()
        assert np.array_equal(r[4], tk.numpy()) and r[5].tobytes() == ts.numpy().tobytes()
        assert r[6] == [(0, 12), (0, 64), (0,)]
