"""DistValueSim: the reference's distributional agent (agents/DistValueSimOnline.py:12-94, BASELINE configs[4]) rebuilt from
its working parts - the file itself does not import (it names `model.ValueSim`, `select_trace_obs_dist`,
`backup_trace_obs_dist`, none of which exist), so there is no behaviour to be a drop-in for beyond the pieces:

  * the tree of TreeAgent WITHOUT the observation projection: per-node statistics (visit, mean, score, variance, M2) and
    a distribution of `atoms` bins over [vmin, vmax) per node, as agents/core_distributional.py's kernels hold them;
  * selection = select_trace_distributional (core_distributional.py:81-104) with the check_low it means to call
    (under-visited children first, low = 5; libc rand()) and policy_dist (:66-79);
  * leaf evaluation = model_distributional.Net (model/model_distributional.py:18-57: softmax over the atoms; input the
    reference's 22 x 10 board = the 20 visible rows under two empty ones) as hand-written HIP kernels (csrc/distnet.hip)
    launched by the native loop of csrc/search.hip; v_dummy for a finished game;
  * backup = backup_trace_distributional (:108-124) with r = the leaf's score;
  * compute_stats / the action = DistValueSimOnline.py:76-98.
The tree loop is tree.hip (TM_KIND_DIST: wave_dist_front / wave_dist_back, one lane per atom in the backup); numerics =
oracle/dist_oracle.c, held to a pure-Python run of the reference functions (tests/golden/ref_distpy.npz)."""
import numpy as np
import torch

from .. import store as st
from .agent import TreeAgent


class DistValueSim(TreeAgent):
    kind = st.KIND_DIST
    low = 5

    def __init__(self, atoms=50, vmin=0, vmax=5000, max_nodes=100000, model=None, evaluator=None, online=False,
                 min_visits_to_store=50, memory_size=500000, memory_growth_rate=5000, **kwargs):
        """online: the reference's online leg (DistValueSimOnline.py:116-170) - a collection stores the freed nodes with at
        least `min_visits_to_store` visits whose seven children have all been visited (its commented store_nodes, default 50)
        as (board, distribution, visits) tuples, train_nodes fits the head on them (memory_size / memory_growth_rate as
        ValueSim's)."""
        kwargs.pop("min_visit", None)
        kwargs.pop("gamma", None)                      # the distributional backup does not discount
        self.atoms, self.vrange = int(atoms), (float(vmin), float(vmax))
        self.evaluator = evaluator
        benchmark = kwargs.get("benchmark", False)
        self.online = bool(online) and not benchmark
        kwargs.setdefault("replay_cap", min(int(max_nodes), 16384) if self.online else 0)
        self.memory_size, self.memory_growth_rate, self.n_trains, self._memory = int(memory_size), int(memory_growth_rate), 0, None
        super().__init__(max_nodes=max_nodes, online=self.online, min_visits_to_store=min_visits_to_store, **kwargs)
        if evaluator is None:
            from ..model_distributional import Model_Dist
            self.model = model if model is not None else Model_Dist(atoms=self.atoms)
            if int(getattr(self.model, "atoms", self.atoms)) != self.atoms:
                # the head's parameter blob is indexed with the store's atom count (tm_distnet_forward_requests): another
                # count reads fc_v out of bounds
                raise ValueError("DistValueSim(atoms=%d) was given a model with %d atoms" % (self.atoms, int(self.model.atoms)))
        else:
            self.model = None

    def _build(self, n_games):
        self.n_games = int(n_games)
        self.n_sub = 1
        kw = dict(self._store_kwargs)
        kw.update(dist_bins=self.atoms, dist_range=self.vrange)
        self.store = st.TreeStore(self.n_games, self.max_nodes, **kw)

    def search_model(self):
        """the HIP head is driven by the native launch loop (search.hip); a Python callable or the torch back end by
        TreeAgent.mcts"""
        return self.model if (self.evaluator is None and self.model.backend == "hip") else False

    @torch.no_grad()
    def evaluate_requests(self):
        """The pending leaves' observations -> distributions over the atoms, into the store's eval_dist."""
        s = self.store
        if self.evaluator is None and self.model.backend == "hip":
            self.model.inference_requests(s)                              # nodes rendered inside the convolution kernel
            return
        states = s.render_eval()                                          # int8 [G, 200]; all zero where nothing is asked
        if self.evaluator is not None:
            d = torch.as_tensor(np.asarray(self.evaluator(states.cpu().numpy().reshape(-1, 20, 10)), np.float32), device=s.device)
        else:
            x = torch.zeros(self.n_games, 1, 22, 10, dtype=torch.float32, device=s.device)
            x[:, 0, 2:, :] = states.view(self.n_games, 20, 10).float()   # the reference's net sees 22 rows (model_distributional.py:27)
            d = self.model.model(x)
        s.t["eval_dist"][:, :self.atoms].copy_(d.reshape(self.n_games, self.atoms))

    def get_value(self, node=None):
        """(mean, variance) of the root's distribution (DistValueSimOnline.py:100-109: mean_variance of node_dist)."""
        s = self.store
        g = torch.arange(self.n_games, device=s.device)
        root = s.t["gs"][:, st.GS["ROOT"]].long() if node is None else torch.as_tensor(np.atleast_1d(node), device=s.device).long()
        d = s.t["node_dist"][g, root, :self.atoms].double()
        delta = (self.vrange[1] - self.vrange[0]) / self.atoms
        centre = (torch.arange(self.atoms, device=s.device, dtype=torch.float64) + 0.5) * delta
        mean = (d * centre).sum(1)
        var = (d * centre * centre).sum(1) - mean * mean
        mean, var = mean.cpu().numpy(), var.cpu().numpy()
        return (mean[0], var[0]) if self.n_games == 1 else (mean, var)

    # ---- online training (DistValueSimOnline.py:143-170): tuples harvested on the device at the collections (tree.hip dist_keep /
    # dist_harvest_store), all-gathered over the ranks, the head fitted on their union with Model_Dist's loss ----
    def train_if_collected(self, every=1, min_tuples=1, **kwargs):
        """as ValueSim.train_if_collected: look on every `every`-th call, fit once the job holds `min_tuples` fresh tuples"""
        from .. import dist as tdist
        if not self.online or self.store is None or self.store.s.replay_cap == 0:
            return None
        self._train_calls = getattr(self, "_train_calls", 0) + 1
        if self._train_calls % max(1, int(every)):
            return None
        if tdist.all_sum(int(self.store.t["replay_count"].sum().item()), self.store.device) < max(1, int(min_tuples)):
            return None
        return self.train_nodes(**kwargs)

    def harvested(self):
        """(packed observations int32 [n,12], distributions float32 [n,64], visits float32 [n]) of this rank since the last
        drain; the device buffers are emptied."""
        s = self.store
        keys, dists, visits = s.replay_dist()
        keys, dists, visits = keys.clone(), dists.clone(), visits.clone()
        s.t["replay_count"].zero_()
        return keys, dists, visits

    def train_nodes(self, dump_data=False, dump_path="./data/memory_dump", **train_kwargs):
        from sys import stderr
        from .. import dist as tdist
        if not self.online or self.evaluator is not None:
            return None
        keys, dists, visits = self.harvested()
        dropped = self.store.counter("N_DROPPED")
        if dropped > getattr(self, "_dropped_seen", 0):
            print("WARNING: {} harvested tuples did not fit the device replay buffer (replay_cap={})".format(
                dropped - getattr(self, "_dropped_seen", 0), self.store.s.replay_cap), file=stderr, flush=True)
            self._dropped_seen = dropped
        if self._memory is not None:
            keys, dists, visits = (torch.cat([a, b]) for a, b in zip(self._memory, (keys, dists, visits)))
        keys, dists, visits = keys[:self.memory_size], dists[:self.memory_size], visits[:self.memory_size]
        keys_all, dists_all, visits_all = tdist.all_gather_rows(keys.view(torch.int32), dists, visits)
        keys_all, dists_all, visits_all = tdist.job_memory(self.memory_size, keys_all, dists_all, visits_all)      # (memory_size is the job's)
        d_size = int(keys_all.shape[0])
        m_size = min(self.n_trains * self.memory_growth_rate, self.memory_size)
        if d_size < max(m_size, 1):
            print("Not enough training data ({} < {}), collecting more data.".format(d_size, m_size), file=stderr, flush=True)
            self._memory = (keys, dists, visits)
            return None
        print("Enough training data ({} >= {}), proceed to training.".format(d_size, m_size), file=stderr, flush=True)
        states = torch.zeros(d_size, 1, 22, 10, dtype=torch.float32, device=keys_all.device)
        states[:, :, 2:, :] = tdist.render_observations(keys_all)          # the net's 22 rows: two empty ones on top
        data = [states, dists_all[:, :self.atoms].contiguous(), visits_all.reshape(-1, 1)]
        if dump_data and tdist.rank() == 0:
            import os
            os.makedirs(os.path.dirname(os.path.abspath(dump_path)), exist_ok=True)
            np.savez(dump_path, states=data[0].cpu().numpy(), values=data[1].cpu().numpy(), weights=data[2].cpu().numpy())
        self.n_trains += 1
        opts = dict(iters_per_val=100, batch_size=1024, max_iters=50000)     # DistValueSimOnline.py:165
        opts.update(train_kwargs)
        res = self.model.train_data(data, **opts)
        self.model.training(False)
        self._memory = None
        print("Training complete.", file=stderr, flush=True)
        return res
