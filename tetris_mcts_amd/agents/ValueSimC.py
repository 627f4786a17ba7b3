"""Batched mirror of the reference's `ValueSimC` (agents/ValueSimC.py:16-45), the wrapper around the all-C++
`OnlineMCTSAgent` / `MCTSAgent` (agents/cppmodule/agent.cpp:392-568): same search as ValueSim(LP) but with the
C++ agent's own numerics — the backed-up value is carried as a float (agent.cpp:496-513), terminal children are
detected through end_obs[o] (agent.cpp:538) and the averaged variance has no gamma^2 (agent.cpp:557-558)."""
from .. import store as st
from .ValueSim import ValueSim


class ValueSimC(ValueSim):
    kind = st.KIND_CPPAGENT_LP

    def __init__(self, sims=100, max_nodes=100000, online=False, accumulation_policy=1, memory_size=10000000,
                 episodes_per_train=25, memory_growth_rate=5000, min_visit=40, projection=True, gamma=0.999,
                 benchmark=False, leaf_parallel=True, **kwargs):
        if not leaf_parallel:
            self.kind = st.KIND_CPPAGENT
        self.accumulation_policy = accumulation_policy
        self.episodes_per_train = episodes_per_train
        super().__init__(sims=sims, max_nodes=max_nodes, online=online, memory_size=memory_size,
                         memory_growth_rate=memory_growth_rate, min_visits_to_store=min_visit, projection=projection,
                         gamma=gamma, benchmark=benchmark, **kwargs)
