"""Batched mirror of the reference's `ValueSimC` (agents/ValueSimC.py:16-45), the wrapper around the all-C++
`OnlineMCTSAgent` / `MCTSAgent` (agents/cppmodule/agent.cpp:392-816): same search as ValueSim(LP) but with the
C++ agent's own numerics — the backed-up value is carried as a float (agent.cpp:496-513), terminal children are
detected through end_obs[o] (agent.cpp:538) and the averaged variance has no gamma^2 (agent.cpp:557-558) — and, when
`online`, the C++ agent's replay memory with its accumulation policies 0-3 (agent.cpp:619-816, `..replay`)."""
from sys import stderr

from .. import store as st
from ..replay import ReplayMemory
from .ValueSim import ValueSim


class ValueSimC(ValueSim):
    kind = st.KIND_CPPAGENT_LP

    def __init__(self, sims=100, max_nodes=100000, online=False, accumulation_policy=1, memory_size=10000000,
                 episodes_per_train=25, memory_growth_rate=5000, min_visit=40, projection=True, gamma=0.999,
                 benchmark=False, leaf_parallel=True, train=None, dump_path="./data/dump", **kwargs):
        """`train`: optional callable(state, value, variance, visit, size) standing in for the `train` argument of
        OnlineMCTSAgent (agent.cpp:845); default = ValueSimC.py:6-14 (np.savez dump + Model.train_data)."""
        if not leaf_parallel:
            self.kind = st.KIND_CPPAGENT
        self.accumulation_policy = accumulation_policy
        self.episodes_per_train = episodes_per_train
        self._train_hook = train
        self.dump_path = dump_path
        super().__init__(sims=sims, max_nodes=max_nodes, online=online, memory_size=memory_size,
                         memory_growth_rate=memory_growth_rate, min_visits_to_store=min_visit, projection=projection,
                         gamma=gamma, benchmark=benchmark, **kwargs)
        self.memory = None
        if online and not benchmark:
            self.memory = ReplayMemory(accumulation_policy, memory_size, episodes_per_train, memory_growth_rate)

    def play(self):
        a = super().play()
        if self.memory is not None:
            # the reference runs store_nodes + the policy inside remove_nodes(), i.e. inside play(); the device GC only
            # harvests, the policy runs here, before the move is applied (same current_episode as in the reference)
            self.train_nodes()
        return a

    def train_nodes(self, **train_kwargs):
        import torch
        from .. import dist as tdist
        s = self.store
        if self.memory is None:
            return None
        harvested = tdist.all_sum(int(s.t["replay_count"].sum().item()), s.device)
        if harvested == 0:
            return None
        print("\nWARNING: REMOVING UNUSED NODES...\nStoring unused nodes...", file=stderr, flush=True)
        keys, stats = s.replay()
        s.t["replay_count"].zero_()
        keys, stats = tdist.all_gather_tuples(keys.view(torch.int32), stats)
        episode = tdist.all_sum(self.episode, s.device)
        out = self.memory.absorb(keys, stats, episode, log=lambda msg: print(msg, file=stderr, flush=True))
        if out is None:
            return None
        state, value, variance, visit = tdist.training_arrays(*out)
        d_size = int(state.shape[0])
        if self._train_hook is not None:
            res = self._train_hook(state, value, variance, visit, d_size)
        elif self.evaluator is None:
            res = self.training(state, value, variance, visit, d_size, **train_kwargs)
        else:
            res = None
        print("Training complete.", file=stderr, flush=True)
        return res

    def training(self, state, value, variance, visit, d_size, **train_kwargs):
        """ValueSimC.py:6-14: dump the training set in the reference's np.savez layout, fit, back to inference mode."""
        if self.dump_path:
            self.dump_training_set(self.dump_path, state[:d_size], value[:d_size], variance[:d_size], visit[:d_size])
        opts = dict(iters_per_val=100, batch_size=512, max_iters=50000, sample_replacement=True, oversampling=False)
        opts.update(train_kwargs)
        res = self.model.train_data([state[:d_size], value[:d_size], variance[:d_size], visit[:d_size]], **opts)
        self.model.training(False)
        return res
