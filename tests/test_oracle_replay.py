"""oracle/replay_oracle.py (restatement of OnlineMCTSAgent's replay memory, agents/cppmodule/agent.cpp:569-816)
against the reference's own compiled OnlineMCTSAgent: tests/golden/ref_online_cpp.json holds every train() payload of
900-move runs under the four accumulation policies (tests/golden/make_golden.py online)."""
import hashlib
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import replay_oracle as ro  # noqa: E402


def test_std_mt19937_known_answer():
    """ISO C++ [rand.predef]: the 10000th consecutive invocation of a default-constructed mt19937 is 4123659995."""
    g = ro.StdMt19937()
    for _ in range(9999):
        g()
    assert g() == 4123659995


def test_std_shuffle_matches_libstdcxx(tmp_path):
    """std::shuffle / uniform_real over std::mt19937(123) compiled here with the reference's compiler (g++, libstdc++)
    vs the restatement: both shuffle regimes (pairwise draws for n^2 < 2^32, one draw per element above)."""
    import subprocess
    src = tmp_path / "s.cpp"
    src.write_text(r'''
#include <algorithm>
#include <cstdio>
#include <numeric>
#include <random>
#include <vector>
int main() {
    std::mt19937 mt(123);
    std::uniform_real_distribution<double> unif(0., 1.);
    for (int n : {1, 2, 7, 400, 70000}) {
        std::vector<int> v(n);
        std::iota(v.begin(), v.end(), 0);
        std::shuffle(v.begin(), v.end(), mt);
        std::shuffle(v.begin(), v.end(), mt);
        unsigned long long h = 1469598103934665603ull;
        for (int x : v) h = (h ^ (unsigned)x) * 1099511628211ull;
        std::printf("%d %llu %.17g\n", n, h, unif(mt));
    }
}''')
    exe = tmp_path / "s"
    subprocess.check_call(["g++", "-O2", "-o", str(exe), str(src)])
    want = subprocess.check_output([str(exe)]).decode().split("\n")
    g = ro.StdMt19937(123)
    for line, n in zip(want, (1, 2, 7, 400, 70000)):
        v = list(range(n))
        ro.std_shuffle(v, g)
        ro.std_shuffle(v, g)
        h = 1469598103934665603
        for x in v:
            h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        assert line == "%d %d %.17g" % (n, h, ro.canonical_double(g))


def _payload(entries):
    st = np.stack([e[0] for e in entries]).astype(np.int8) if entries else np.zeros((0, 200), np.int8)
    f = lambda k: np.asarray([e[k] for e in entries], "<f4").tobytes().hex()
    return dict(size=len(entries), states_sha1=hashlib.sha1(st.tobytes()).hexdigest(), value=f(1), variance=f(2),
                visit=f(3))


@pytest.mark.parametrize("idx", [0, 1, 2, 3])
def test_online_memory_matches_reference(oracle, golden_dir, idx):
    with open(os.path.join(golden_dir, "ref_online_cpp.json")) as f:
        r = json.load(f)[idx]
    mem = ro.OnlineMemory(r["policy"], r["memory_size"], r["episodes_per_train"], r["growth"])
    g = oracle.Game(seed=r["seed"])
    a = oracle.Agent(2, max_nodes=r["max_nodes"], online=True, min_visits_to_store=r["min_visit"], memory_size=1 << 20,
                     cpp_occupied=True)
    a.update_root(g)
    calls, seen, n_gc = [], 0, 0
    for m, act in enumerate(r["actions"]):
        got = a.play(r["sims"])
        assert got == act, (m, got, act)
        if a.n_gc != n_gc:
            assert a.n_gc == n_gc + 1     # one GC per move at most: the policy step is per GC
            n_gc = a.n_gc
            st, val, var, vis = a.memory()
            batch = [(st[i], val[i], var[i], vis[i]) for i in range(seen, len(val))]
            seen = len(val)
            out = mem.remove_nodes(batch, a.episode)
            if out is not None:
                calls.append(dict(_payload(out), move=m))
        g.play(got)
        a.update_root(g)
        if g.end:
            g.reset()
            a.update_root(g)
    assert len(calls) == len(r["train_calls"])
    for c, w in zip(calls, r["train_calls"]):
        assert c == w, (c["move"], c["size"], w["move"], w["size"])
    a.close()
