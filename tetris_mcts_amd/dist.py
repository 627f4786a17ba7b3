"""Multi-GPU: one process per GPU, games sharded by index, one exchange step.

The reference has no collective at all (replicas write separate files, cycle.sh:69-73).  Games are independent
(every agent owns its tree, agents/agent.py:58-88), so rank r simply owns games [r*G/P, (r+1)*G/P) with private
pools and RNG streams.  The only exchange is the all-gather of the (state, TD-target) training tuples that GC
harvests (ValueSim.store_nodes, ValueSim.py:122-159), so every rank can train the same replica on the union
(train.train_data then splits each batch over the ranks and averages the gradients with one all-reduce per iteration).
Backend: "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
"""
import math

import torch
import torch.distributed as dist


def shard_range(n_total, rank, world):
    """Contiguous, balanced shard: (start, count) of rank's games out of n_total."""
    base, rem = divmod(int(n_total), int(world))
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


def game_seeds(base_seed, games_per_rank, rank, world=None, start=None):
    """Environment seeds of one rank's games: game g of rank r is game r * games_per_rank + g of the job, seeded
    base_seed + that index - a game's whole trajectory depends on its index in the job only, never on how many ranks
    the job has or on which of them it runs (tests/test_dist.py).  `games_per_rank` is the shard size of EVERY rank (equal
    shards, what bench.py and the agents use); with the uneven shards of shard_range() pass the shard's `start` (its first
    game's index in the job) instead, so that adjacent ranks do not overlap."""
    import numpy as np
    first = int(rank) * int(games_per_rank) if start is None else int(start)
    return int(base_seed) + first + np.arange(int(games_per_rank), dtype=np.int64)


def rank(group=None):
    return dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0


def all_sum(value, device="cpu", group=None):
    """Sum of a Python integer over ranks (episode counters, harvest counts)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64, device=device)
    dist.all_reduce(t, group=group)
    return int(t.item())


def _gather_device(t, group=None):
    """Where a payload lives while it is exchanged: the tensor's own device with RCCL; host memory when the group's backend is
    gloo and the tensor is on a GPU (gloo's all_gather takes no device tensors - two ranks sharing one GPU in the tests, CPU-only
    jobs).  The result goes back to the tensor's device."""
    if t.device.type == "cuda" and dist.get_backend(group) == "gloo":
        return torch.device("cpu")
    return t.device


def all_gather_tuples(obs_keys, stats, group=None):
    """All-gather variable-length training tuples.

    obs_keys: int32 [n, 12] packed observations (ENGINE_SPEC.md section 7); stats: float32 [n, 4] = value, variance,
    visit, 0.  Returns (obs_keys_all, stats_all) = the concatenation over ranks in rank order (identical on every
    rank).  One count all-gather (8 B/rank) + one padded payload all-gather (64 B/tuple).
    """
    if not (dist.is_available() and dist.is_initialized()):
        return obs_keys, stats
    world = dist.get_world_size(group)       # (a one-rank group still goes through the collective: same code path as N ranks)
    home, dev = obs_keys.device, _gather_device(obs_keys, group)
    n = torch.tensor([obs_keys.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    m = max(counts)
    if m == 0:
        return obs_keys, stats
    # one fused payload: 12 key words + 4 stat words (bit-cast) per tuple
    pay = torch.zeros(m, 16, dtype=torch.int32, device=home)
    pay[:obs_keys.shape[0], :12] = obs_keys.to(torch.int32)
    pay[:obs_keys.shape[0], 12:] = stats.contiguous().view(torch.int32)
    pay = pay.to(dev)
    out = [torch.empty_like(pay) for _ in range(world)]
    dist.all_gather(out, pay, group=group)
    allp = torch.cat([out[r][:counts[r]] for r in range(world)], dim=0).to(home)
    return allp[:, :12].contiguous(), allp[:, 12:].contiguous().view(torch.float32)


def all_gather_rows(*cols, group=None):
    """All-gather variable-length rows made of several 4-byte-element tensors with the same first dimension (the
    distributional agent's tuples: packed observation int32 [n,12], distribution float32 [n,64], visits float32 [n]): one count
    all-gather + one padded payload all-gather of the bit-cast columns; returns the columns concatenated over ranks in rank
    order, with their dtypes and trailing shapes."""
    if not (dist.is_available() and dist.is_initialized()):
        return cols
    world = dist.get_world_size(group)
    home, dev = cols[0].device, _gather_device(cols[0], group)
    n_loc = cols[0].shape[0]
    # explicit widths: a rank that has not harvested yet has 0 rows (collections are not synchronised over the ranks), and
    # reshape(0, -1) is refused - it must still take part in both collectives, or the others hang
    if any(c.element_size() != 4 for c in cols):
        raise TypeError("all_gather_rows: 4-byte elements only (the columns travel bit-cast to int32): " + ", ".join(str(c.dtype) for c in cols))
    widths = [max(1, math.prod(c.shape[1:])) for c in cols]
    flat = [c.reshape(n_loc, w).contiguous().view(torch.int32) for c, w in zip(cols, widths)]
    n = torch.tensor([n_loc], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    m = max(counts)
    if m == 0:
        return cols
    pay = torch.zeros(m, sum(widths), dtype=torch.int32, device=home)
    pay[:n_loc] = torch.cat(flat, dim=1)
    pay = pay.to(dev)
    out = [torch.empty_like(pay) for _ in range(world)]
    dist.all_gather(out, pay, group=group)
    allp = torch.cat([out[r][:counts[r]] for r in range(world)], dim=0).to(home)
    res, off = [], 0
    for c, w in zip(cols, widths):
        res.append(allp[:, off:off + w].contiguous().view(c.dtype).reshape((allp.shape[0],) + tuple(c.shape[1:])))
        off += w
    return tuple(res)


def job_memory(memory_size, *cols):
    """The all-gathered training tuples cut to the JOB's replay memory: `memory_size` is what the reference's one process holds
    for all its games (agents/ValueSim.py:14, 122-159: store_nodes stops at memory_size), so P ranks together keep the first
    memory_size tuples of the gathered set - rank after rank, each rank's tuples in the order of its harvest: the same on every
    rank - not P times that."""
    return tuple(c[:memory_size] for c in cols)


def render_observations(obs_keys):
    """Packed observations int32 [n,12] -> float32 [n,1,20,10] network inputs (0 empty, 1 locked, -1 falling piece);
    the layout of the reference's replay memory (ValueSim.py:24-29).  Plain tensor ops (off the hot path)."""
    k = obs_keys.to(torch.int64) & 0xFFFFFFFF
    n = k.shape[0]
    rows = torch.stack([(k[:, r // 2] >> (16 * (r % 2))) & 0xFFFF for r in range(20)], dim=1)      # [n,20]
    cols = torch.arange(10, device=k.device)
    board = ((rows[:, :, None] >> cols[None, None, :]) & 1).to(torch.float32)                       # [n,20,10]
    cells = torch.stack([(k[:, 10] >> (8 * i)) & 0xFF for i in range(4)], dim=1)                    # [n,4]
    ended = (k[:, 11] & 0xFF) != 0
    flat = board.reshape(n, 200)
    idx = cells.clamp(max=199)
    piece = torch.zeros_like(flat).scatter_(1, idx, 1.0) * (~ended)[:, None].to(torch.float32)
    piece = piece * (cells.min(dim=1).values < 200)[:, None].to(torch.float32)
    flat = torch.where(piece > 0, torch.full_like(flat, -1.0), flat)
    return flat.reshape(n, 1, 20, 10)


def training_arrays(obs_keys, stats):
    """(states [n,1,20,10], values [n,1], variance [n,1], weights [n,1]) as ValueSim.memory holds them."""
    return render_observations(obs_keys), stats[:, 0:1].clone(), stats[:, 1:2].clone(), stats[:, 2:3].clone()
