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
        # device-side harvest buffer per game (64 B per tuple); a GC at a 100 000-entry pool frees a few thousand
        # observations with enough visits, and whatever exceeds the buffer between two drains is dropped
        kwargs.setdefault("replay_cap", min(int(max_nodes), 16384) if (online and not benchmark) else 0)
        super().__init__(max_nodes=max_nodes, gamma=gamma, online=(online and not benchmark),
                         min_visits_to_store=min_visits_to_store, **kwargs)
        self.online = online
        self.n_trains = 0
        self._memory = None       # (packed observations, stats) carried over between trainings
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

    def search_model(self):
        return self.model if (self.evaluator is None and self.model.backend == "hip") else False

    def evaluate_requests(self):
        if self.evaluator is None and self.model.backend == "hip":
            self.model.inference_requests(self.store)   # observations rendered inside the conv kernel
        else:
            super().evaluate_requests()

    # ---- online training (ValueSim.py:161-185): the reference trains inside remove_nodes(), in the middle of a
    # simulation of its single game; the batched engine harvests tuples at GC time on the device and trains between
    # moves, on the union over all games and all ranks (tetris_mcts_amd/dist.py) ----
    @staticmethod
    def dump_training_set(path, state, value, variance, visit):
        """np.savez layout of the reference's replay dump (ValueSim.py:176-177, ValueSimC.py:8): keys states [n,1,20,10],
        values / variance / weights [n,1]; written by rank 0 only."""
        import os
        import numpy as np
        from .. import dist as tdist
        if tdist.rank() != 0:
            return
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        np.savez(path, states=state.cpu().numpy(), values=value.cpu().numpy(), variance=variance.cpu().numpy(),
                 weights=visit.cpu().numpy())

    def train_if_collected(self, every=1, min_tuples=1, **kwargs):
        """train_nodes() if garbage collections have harvested tuples since the last fit (the reference's remove_nodes ->
        store_nodes -> train_nodes chain, ValueSim.py:101-120, runs at every collection of its one game); None otherwise.
        One game: call it after every move with the defaults and the cadence is the reference's.  A batch of games collects
        somewhere on nearly every move, and looking costs a device sync and a collective: `every` = look only on every k-th
        call, `min_tuples` = fit only once the job (all ranks) holds that many fresh tuples (play.py: --train_every,
        --train_min_tuples)."""
        from .. import dist as tdist
        if not self.online or self.store is None or self.store.s.replay_cap == 0:
            return None
        self._train_calls = getattr(self, "_train_calls", 0) + 1
        if self._train_calls % max(1, int(every)):
            return None
        # (what an earlier look moved into the host-side memory - "not enough training data" - counts: it is part of the next
        # fit's set.  Every rank must call this on the same moves: the count is a collective)
        held = 0 if self._memory is None else int(self._memory[0].shape[0])
        if tdist.all_sum(int(self.store.t["replay_count"].sum().item()) + held, self.store.device) < max(1, int(min_tuples)):
            return None
        return self.train_nodes(**kwargs)

    def train_nodes(self, dump_data=False, dump_path="./data/dump", **train_kwargs):
        import torch
        from .. import dist as tdist
        from sys import stderr
        if not self.online or self.evaluator is not None:
            return None
        s = self.store
        keys, stats = s.replay()
        s.t["replay_count"].zero_()
        dropped = s.counter("N_DROPPED")
        if dropped > getattr(self, "_dropped_seen", 0):
            # the reference keeps every qualifying observation up to memory_size (ValueSim.py:122-159): say so when the
            # device-side harvest buffer was too small between two drains
            print("WARNING: {} harvested tuples did not fit the device replay buffer (replay_cap={}); drain more often or "
                  "raise replay_cap".format(dropped - getattr(self, "_dropped_seen", 0), s.s.replay_cap), file=stderr, flush=True)
            self._dropped_seen = dropped
        if self._memory is not None:
            keys = torch.cat([self._memory[0], keys])
            stats = torch.cat([self._memory[1], stats])
        keys, stats = keys[:self.memory_size], stats[:self.memory_size]
        keys_all, stats_all = tdist.all_gather_tuples(keys.view(torch.int32), stats)
        keys_all, stats_all = tdist.job_memory(self.memory_size, keys_all, stats_all)      # (memory_size is the job's, not a rank's)
        d_size = keys_all.shape[0]
        m_size = min(self.n_trains * self.memory_growth_rate, self.memory_size)
        if d_size < max(m_size, 1):
            print("Not enough training data ({} < {}), collecting more data.".format(d_size, m_size), file=stderr, flush=True)
            self._memory = (keys, stats)
            return None
        print("Enough training data ({} >= {}), proceed to training.".format(d_size, m_size), file=stderr, flush=True)
        data = list(tdist.training_arrays(keys_all, stats_all))
        if dump_data:
            self.dump_training_set(dump_path, *data)
        self.n_trains += 1
        opts = dict(iters_per_val=100, batch_size=1024, max_iters=50000)
        opts.update(train_kwargs)
        res = self.model.train_data(data, **opts)
        self.model.training(False)
        self._memory = None
        print("Training complete.", file=stderr, flush=True)
        return res
