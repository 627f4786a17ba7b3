"""Batched mirror of the reference's `VanillaC` (agents/VanillaC.py:5-17): the all-C++ `MCTSAgent`
(agents/cppmodule/agent.cpp:392-460) with evaluator type 1 - after expanding the leaf, a copy of it is played to the end
with `random.randint(0, 7)` actions (7 is no action: only gravity acts), value = final score, variance 1e5 - and the
C++ agent's numerics: check_low threshold 1, gamma 0.99, the backed-up value carried as a float (agent.cpp:496-513),
root statistics value + score[c] - score[root] with std::max_element (agent.cpp:149-172).
The playout RNG is CPython's Mersenne Twister, one stream per game, seeded as in agents.Vanilla."""
from .. import store as st
from .Vanilla import Vanilla


class VanillaC(Vanilla):
    kind = st.KIND_VANILLA_C
    low = 1

    def __init__(self, sims=100, max_nodes=500000, projection=True, gamma=0.99, benchmark=False, LP=False, **kwargs):
        super().__init__(sims=sims, max_nodes=max_nodes, gamma=gamma, benchmark=benchmark, **kwargs)
