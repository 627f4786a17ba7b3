"""Replay memory of the reference's all-C++ online agent, `OnlineMCTSAgent` (agents/cppmodule/agent.cpp:569-816):
harvested (observation, value, variance, visit) tuples are accumulated until an *accumulation policy* says "train":

    0  train every `episodes_per_train` episodes; tuples are dropped at random so that about one memory's worth
       arrives per training period; a full memory loses a random 1 % (agent.cpp:635-661, 751-775)
    1  train every `episodes_per_train` episodes; a full memory loses its least-visited percentile
       (agent.cpp:662-675, 710-749)
    2  train every `episodes_per_train` episodes or as soon as the memory is full (agent.cpp:676-686)
    3  train once `min(n_trains * memory_growth_rate, memory_size)` tuples are there (agent.cpp:687-696), which is also
       the rule of the Python ValueSim (ValueSim.py:161-185)

The tuples live where they were harvested (device tensors: 12-dword packed observations + float stats), the policy
is host arithmetic on a handful of scalars.  The random parts draw from `std::mt19937 mt(123)` exactly as libstdc++
does (agent.cpp:29-30: generate_canonical over two 32-bit draws; std::shuffle with Lemire's bounded integers), so a
single-game agent reproduces the reference's training sets bit for bit; with many games the harvests of all games (and
ranks) arrive concatenated in game order and `current_episode` is the total number of finished episodes.
"""
import math

import numpy as np
import torch


class StdMt19937:
    """std::mt19937 seeded with an integer: MT19937 with init_genrand seeding, which is also numpy's legacy seeding."""

    def __init__(self, seed=123):
        legacy = np.random.RandomState(seed).get_state()
        self.bg = np.random.MT19937()
        self.bg.state = {"bit_generator": "MT19937", "state": {"key": legacy[1], "pos": int(legacy[2])}}

    def raw(self, n):
        return self.bg.random_raw(int(n))

    def next(self):
        return int(self.bg.random_raw())

    def canonical(self, n):
        """n draws of std::uniform_real_distribution<double>(0, 1): (lo + hi * 2^32) / 2^64, two engine calls each."""
        r = self.raw(2 * n).astype(np.float64)
        u = (r[0::2] + r[1::2] * 4294967296.0) / 18446744073709551616.0
        return np.where(u >= 1.0, math.nextafter(1.0, 0.0), u)

    def bounded(self, rng):
        """std::uniform_int_distribution<size_t>{0, rng - 1} on a 32-bit engine (multiply-shift with rejection)."""
        product = self.next() * rng
        low = product & 0xFFFFFFFF
        if low < rng:
            threshold = (0x100000000 - rng) % rng
            while low < threshold:
                product = self.next() * rng
                low = product & 0xFFFFFFFF
        return product >> 32

    def shuffle(self, v):
        """std::shuffle(v.begin(), v.end(), mt) of libstdc++ on a Python list: two swap positions per draw while
        len(v)^2 fits the engine's range, one draw per element beyond."""
        n = len(v)
        if n < 2:
            return
        if 0xFFFFFFFF // n >= n:
            i = 1
            if n % 2 == 0:
                j = self.bounded(2)
                v[1], v[j] = v[j], v[1]
                i = 2
            while i < n:
                x = self.bounded((i + 1) * (i + 2))
                a, b = divmod(x, i + 2)
                v[i], v[a] = v[a], v[i]
                v[i + 1], v[b] = v[b], v[i + 1]
                i += 2
        else:
            for i in range(1, n):
                j = self.bounded(i + 1)
                v[i], v[j] = v[j], v[i]


class ReplayMemory:
    COL_VALUE, COL_VARIANCE, COL_VISIT = 0, 1, 2       # columns of `stats` (tm_store.replay_stat)

    def __init__(self, accumulation_policy=1, memory_size=10000000, episodes_per_train=25, memory_growth_rate=5000,
                 seed=123):
        if accumulation_policy not in (0, 1, 2, 3):
            raise ValueError("accumulation_policy must be 0..3")
        if accumulation_policy == 0 and int(memory_size * 0.01) < 1:
            raise ValueError("policy 0 trims int(memory_size * 0.01) entries: memory_size must be >= 100")
        self.policy = accumulation_policy
        self.memory_size = int(memory_size)
        self.episodes_per_train = int(episodes_per_train)
        self.memory_growth_rate = int(memory_growth_rate)
        self.keys = None
        self.stats = None
        self.nodes_per_episode = []
        self.accumulated_nodes = 0
        self.last_accumulation_episode = 0
        self.last_training_episode = 0
        self.memory_drop_prob = 0.0
        self.n_trains = 0
        self.mt = StdMt19937(seed)
        self._sampler = None

    @property
    def memory_index(self):
        return 0 if self.keys is None else int(self.keys.shape[0])

    def _append(self, keys, stats):
        if self.keys is None:
            self.keys, self.stats = keys.clone(), stats.clone()
        else:
            self.keys, self.stats = torch.cat([self.keys, keys]), torch.cat([self.stats, stats])

    # ---- store_nodes (agent.cpp:777-816); the min_visit / end filter already ran on the device, at GC ----
    def store(self, keys, stats):
        n, space = int(keys.shape[0]), self.memory_size - self.memory_index
        if n == 0 or space <= 0:      # (a full memory at entry cannot happen in the reference: every policy empties or trims it)
            return
        if self.policy != 0:
            take = min(n, space)          # the reference's loop stops at the tuple that fills the memory
            self.accumulated_nodes += take
            self._append(keys[:take], stats[:take])
            return
        # policy 0: one uniform draw per visited tuple, the loop stops at the tuple that fills the memory
        before = self.mt.bg.state
        keep = self.mt.canonical(n) >= self.memory_drop_prob
        filled = np.flatnonzero(np.cumsum(keep) == space) if space <= int(keep.sum()) else ()
        used = n
        if len(filled):
            used = int(filled[0]) + 1
            self.mt.bg.state = before
            keep = (self.mt.canonical(used) >= self.memory_drop_prob)
        self.accumulated_nodes += used
        sel = torch.from_numpy(np.flatnonzero(keep[:used])).to(keys.device)
        self._append(keys[:used][sel], stats[:used][sel])

    def _weighted_trimming(self, percentile):
        """agent.cpp:710-749.  Entries whose visit count is <= the percentile's are squeezed out; the reference's
        loop does not count the first one, so the slot just behind the survivors keeps its old content and stays in
        the memory - kept, it is part of what the reference trains on."""
        vis = self.stats[:, self.COL_VISIT]
        threshold = torch.sort(vis.to(torch.int32)).values[int(self.memory_size * percentile)].to(vis.dtype)
        drop = vis <= threshold
        n_keep = int((~drop).sum())
        tail = slice(n_keep, n_keep + 1)
        self.keys = torch.cat([self.keys[~drop], self.keys[tail]])
        self.stats = torch.cat([self.stats[~drop], self.stats[tail]])

    def _random_trimming(self, fraction):
        """agent.cpp:751-775 with the persistent IntSampler of agent.cpp:51-64."""
        if self._sampler is None:
            self._sampler = list(range(self.memory_size))
        self.mt.shuffle(self._sampler)
        gone = torch.tensor(self._sampler[:int(self.memory_size * fraction)], dtype=torch.long, device=self.keys.device)
        keep = torch.ones(self.memory_index, dtype=torch.bool, device=self.keys.device)
        keep[gone] = False
        self.keys, self.stats = self.keys[keep], self.stats[keep]

    # ---- the online half of remove_nodes (agent.cpp:619-708) ----
    def absorb(self, keys, stats, current_episode, log=None):
        """Add one harvest and apply the accumulation policy.  Returns (keys, stats) to train on - the memory is then
        empty again - or None while collecting."""
        say = log if log is not None else (lambda *_: None)
        self.store(keys, stats)
        say("Memory usage: {} / {}".format(self.memory_index, self.memory_size))
        diff = current_episode - self.last_training_episode
        full = self.memory_index >= self.memory_size
        if self.policy == 0:
            if self.last_accumulation_episode != current_episode:
                self.nodes_per_episode.append(self.accumulated_nodes)
                if len(self.nodes_per_episode) > self.episodes_per_train:
                    self.nodes_per_episode.pop(0)
                total = sum(self.nodes_per_episode)
                self.memory_drop_prob = max(0.0, 1.0 - float(self.memory_size) / total) if total else 0.0
                self.accumulated_nodes = 0
                self.last_accumulation_episode = current_episode
                say("Average nodes stored per episode: {}    Memory dropping probability: {:g}".format(
                    total // len(self.nodes_per_episode), self.memory_drop_prob))
            passed = diff >= self.episodes_per_train
        elif self.policy == 1:
            passed = diff >= self.episodes_per_train
        elif self.policy == 2:
            passed = diff >= self.episodes_per_train or full
        else:
            m_size = min(self.n_trains * self.memory_growth_rate, self.memory_size)
            passed = self.memory_index >= m_size
            say(("Enough training data ({} >= {}), proceed to training." if passed else
                 "Not enough training data ({} < {}), collecting more data.").format(self.memory_index, m_size))
        if self.policy in (0, 1, 2):
            if diff >= self.episodes_per_train:
                say("Enough episodes ({} >= {}), proceed to training.".format(diff, self.episodes_per_train))
            elif self.policy == 2 and full:
                say("Memory limit exceeded ({} >= {}), proceed to training.".format(self.memory_index, self.memory_size))
            else:
                if full and self.policy != 2:
                    say("Memory limit exceeded, trimming memory.")
                    if self.policy == 0:
                        self._random_trimming(0.01)
                    else:
                        self._weighted_trimming(0.01)
                    say("Memory usage: {} / {}".format(self.memory_index, self.memory_size))
                say("Not enough episodes ({} < {}), collecting more episodes.".format(diff, self.episodes_per_train))
        if not passed:
            return None
        out = (self.keys, self.stats)
        if out[0] is None:
            out = (keys[:0], stats[:0])
        self.keys = self.stats = None
        self.n_trains += 1
        self.last_training_episode = current_episode
        return out
