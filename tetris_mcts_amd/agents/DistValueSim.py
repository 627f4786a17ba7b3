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

    def __init__(self, atoms=50, vmin=0, vmax=5000, max_nodes=100000, model=None, evaluator=None, online=False, **kwargs):
        kwargs.pop("min_visit", None)
        kwargs.pop("gamma", None)                      # the distributional backup does not discount
        self.atoms, self.vrange = int(atoms), (float(vmin), float(vmax))
        self.evaluator = evaluator
        if online:
            from sys import stderr
            print("DistValueSim: online training of the distributional head is not built (the reference's own train_nodes "
                  "reads a memory its store_nodes - commented out - never fills); running with online=False", file=stderr)
        super().__init__(max_nodes=max_nodes, online=False, **kwargs)
        if evaluator is None:
            from ..model_distributional import Model_Dist
            self.model = model if model is not None else Model_Dist(atoms=self.atoms)
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
