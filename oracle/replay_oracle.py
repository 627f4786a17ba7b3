"""TEST INFRASTRUCTURE ONLY (see oracle/README in DESIGN.md section 6): CPU restatement, in plain Python loops, of the
replay memory of the reference's all-C++ online agent:

    OnlineMCTSAgent::remove_nodes        agents/cppmodule/agent.cpp:619-708   (accumulation policies 0..3)
    OnlineMCTSAgent::weighted_trimming   agents/cppmodule/agent.cpp:710-749
    OnlineMCTSAgent::random_trimming     agents/cppmodule/agent.cpp:751-775
    OnlineMCTSAgent::store_nodes         agents/cppmodule/agent.cpp:777-816
    IntSampler                           agents/cppmodule/agent.cpp:51-64
    mt / unif                            agents/cppmodule/agent.cpp:29-30     (std::mt19937 mt(123), uniform_real<double>)

The random parts go through libstdc++ (GCC 11.4, the compiler the reference is built with here), whose published
algorithms are restated below: mt19937 (MT19937 with init_genrand seeding), generate_canonical<double,53> (two 32-bit
draws, bits/random.tcc), uniform_int_distribution with a 32-bit engine (Lemire's multiply-shift with rejection,
bits/uniform_int_dist.h:243-317) and std::shuffle (pairwise swaps from one draw while range^2 fits in 32 bits, otherwise
one draw per element, bits/stl_algo.h:3726-3792).

PINNED: tests/test_oracle_replay.py feeds this class the harvests of the oracle agent (itself pinned on the reference's
MCTSAgent) and compares every train() payload with tests/golden/ref_online_cpp.json, recorded from the reference's own
compiled OnlineMCTSAgent (tests/golden/make_golden.py online).
"""


class StdMt19937:
    """std::mt19937 (result_type uint_fast32_t, 32 significant bits)."""

    def __init__(self, seed=5489):
        mt = [0] * 624
        mt[0] = seed & 0xFFFFFFFF
        for i in range(1, 624):
            mt[i] = (1812433253 * (mt[i - 1] ^ (mt[i - 1] >> 30)) + i) & 0xFFFFFFFF
        self.mt, self.pos = mt, 624

    def __call__(self):
        mt = self.mt
        if self.pos >= 624:
            for k in range(624):
                y = (mt[k] & 0x80000000) | (mt[(k + 1) % 624] & 0x7FFFFFFF)
                mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ (0x9908B0DF if y & 1 else 0)
            self.pos = 0
        y = mt[self.pos]
        self.pos += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y


def canonical_double(g):
    """std::uniform_real_distribution<double>(0,1)(g) = generate_canonical<double, 53>: k = 2 draws of 32 bits."""
    s = float(g()) + float(g()) * 4294967296.0
    r = s / 18446744073709551616.0
    if r >= 1.0:
        import math
        r = math.nextafter(1.0, 0.0)
    return r


def uniform_int(g, a, b):
    """std::uniform_int_distribution<unsigned long>{a, b}(g) with a 32-bit engine (b - a < 2**32 - 1)."""
    rng = b - a + 1
    assert 0 < rng < 0xFFFFFFFF
    product = g() * rng
    low = product & 0xFFFFFFFF
    if low < rng:
        threshold = ((1 << 32) - rng) % rng
        while low < threshold:
            product = g() * rng
            low = product & 0xFFFFFFFF
    return a + (product >> 32)


def std_shuffle(v, g):
    n = len(v)
    if n == 0:
        return
    if 0xFFFFFFFF // n >= n:
        i = 1
        if n % 2 == 0:
            j = uniform_int(g, 0, 1)
            v[i], v[j] = v[j], v[i]
            i += 1
        while i != n:
            swap_range = i + 1
            x = uniform_int(g, 0, swap_range * (swap_range + 1) - 1)
            p0, p1 = x // (swap_range + 1), x % (swap_range + 1)
            v[i], v[p0] = v[p0], v[i]
            i += 1
            v[i], v[p1] = v[p1], v[i]
            i += 1
        return
    for i in range(1, n):
        j = uniform_int(g, 0, i)
        v[i], v[j] = v[j], v[i]


class OnlineMemory:
    """The m_state/m_value/m_variance/m_visit memory of OnlineMCTSAgent and its bookkeeping.  Entries are opaque
    Python tuples (state, value, variance, visit); `visit` (last field) drives weighted trimming."""

    def __init__(self, accumulation_policy=0, memory_size=500000, episodes_per_train=25, memory_growth_rate=5000,
                 seed=123):
        self.policy = accumulation_policy
        self.memory_size = memory_size
        self.episodes_per_train = episodes_per_train
        self.memory_growth_rate = memory_growth_rate
        self.mem = [None] * memory_size
        self.memory_index = 0
        self.nodes_per_episode = []
        self.accumulated_nodes = 0
        self.last_accumulation_episode = 0
        self.last_training_episode = 0
        self.memory_drop_prob = 0.0
        self.n_trains = 0
        self.g = StdMt19937(seed)            # agent.cpp:29 (process-wide in the reference)
        self.sampler = None                  # static IntSampler of random_trimming (agent.cpp:752)

    def store_nodes(self, tuples):
        """agent.cpp:777-816; `tuples` = the harvest in available_obs order, already filtered by min_visit / end."""
        for t in tuples:
            self.accumulated_nodes += 1
            if self.policy == 0 and canonical_double(self.g) < self.memory_drop_prob:
                continue
            self.mem[self.memory_index] = t
            self.memory_index += 1
            if self.memory_index == self.memory_size:
                break

    def weighted_trimming(self, percentile):
        M = self.memory_size
        weights = sorted(int(e[3]) for e in self.mem)
        threshold = weights[int(M * percentile)]
        idx_fill = -1
        for i in range(M):
            if self.mem[i][3] <= threshold:
                idx_fill = i
                break
        for i in range(idx_fill + 1, M):
            if self.mem[i][3] <= threshold:
                self.memory_index -= 1
                continue
            self.mem[idx_fill] = self.mem[i]
            idx_fill += 1

    def random_trimming(self, fraction):
        M = self.memory_size
        if self.sampler is None:
            self.sampler = list(range(M))
        std_shuffle(self.sampler, self.g)
        indices = sorted(self.sampler[:int(M * fraction)])
        idx_fill = indices[0]
        for k, it in enumerate(indices):
            end = M if k + 1 == len(indices) else indices[k + 1]
            for start in range(it + 1, end):
                self.mem[idx_fill] = self.mem[start]
                idx_fill += 1
        self.memory_index -= len(indices)

    def remove_nodes(self, tuples, current_episode):
        """agent.cpp:619-708 minus the tree part; returns the list handed to train() or None."""
        self.store_nodes(tuples)
        p, passed = self.policy, False
        diff = current_episode - self.last_training_episode
        if p == 0:
            if self.last_accumulation_episode != current_episode:
                self.nodes_per_episode.append(self.accumulated_nodes)
                if len(self.nodes_per_episode) > self.episodes_per_train:
                    self.nodes_per_episode.pop(0)
                s = sum(self.nodes_per_episode)
                self.memory_drop_prob = max(0.0, 1.0 - float(self.memory_size) / s) if s else 0.0
                self.accumulated_nodes = 0
                self.last_accumulation_episode = current_episode
            passed = diff >= self.episodes_per_train
            if not passed and self.memory_index >= self.memory_size:
                self.random_trimming(0.01)
        elif p == 1:
            passed = diff >= self.episodes_per_train
            if not passed and self.memory_index >= self.memory_size:
                self.weighted_trimming(0.01)
        elif p == 2:
            passed = diff >= self.episodes_per_train or self.memory_index >= self.memory_size
        elif p == 3:
            passed = self.memory_index >= min(self.n_trains * self.memory_growth_rate, self.memory_size)
        if not passed:
            return None
        out = self.mem[:self.memory_index]
        self.n_trains += 1
        self.memory_index = 0
        self.last_training_episode = current_episode
        return out
