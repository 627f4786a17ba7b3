"""Batched mirror of the reference's `ValueSim` agent (agents/ValueSim.py:12-185): per simulation
select -> evaluate the leaf with the value net -> expand -> backup (gamma 0.999, check_low threshold 1,
pool of 100000 nodes per game).  The tree loop runs in tree.hip; leaf states of all games are evaluated
as one batch."""
from .. import store as st
from ..model import Model_VV as Model
from .agent import TreeAgent


class ValueSim(TreeAgent):
    kind = st.KIND_VALUESIM
    low = 1

    def __init__(self, online=True, memory_size=500000, min_visits_to_store=10, gamma=0.999, memory_growth_rate=5000,
                 max_nodes=100000, model=None, evaluator=None, **kwargs):
        kwargs.pop("min_visit", None)  # play.py:89 forwards it; the reference ValueSim ignores it too
        benchmark = kwargs.get("benchmark", False)
        super().__init__(max_nodes=max_nodes, gamma=gamma, online=(online and not benchmark),
                         min_visits_to_store=min_visits_to_store, **kwargs)
        self.online = online
        self.memory_size = memory_size
        self.memory_growth_rate = memory_growth_rate
        self.min_visits_to_store = min_visits_to_store
        self.evaluator = evaluator
        if evaluator is None:
            self.model = model if model is not None else Model()
            if model is None:
                self.model.load()
            self.model.training(False)

    def evaluate(self, states, v_out, var_out):
        if self.evaluator is not None:
            v, var = self.evaluator(states)
            v_out.copy_(v)
            var_out.copy_(var)
        else:
            self.model.inference_device(states, v_out, var_out)

    def evaluate_requests(self):
        if self.evaluator is None and self.model.backend == "hip":
            self.model.inference_requests(self.store)   # observations rendered inside the conv kernel
        else:
            super().evaluate_requests()
