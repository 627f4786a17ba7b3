"""Batched mirror of the reference's `Vanilla` agent (agents/Vanilla.py:8-64): plain UCT with random rollouts —
no network.  Per simulation: select (check_low threshold 5), play a copy of the leaf to the end of the game with
`random.randint(0, 6)` actions, value = final score, variance 1e3, expand, backup with gamma 0.99.
The rollout RNG is CPython's Mersenne Twister, one stream per game: `random_seed=s` gives game g the stream of
`random.seed(s + g)`; with `random_seed=None` and a single game the process-global `random` state is used, as the
reference does (BASELINE configs[0] seeds it with `random.seed(0)`)."""
import random

from .. import store as st
from .agent import TreeAgent


class Vanilla(TreeAgent):
    kind = st.KIND_VANILLA
    low = 5

    def __init__(self, gamma=0.99, max_nodes=500000, random_seed=None, **kwargs):
        kwargs.pop("min_visit", None)
        kwargs.pop("online", None)
        self.random_seed = random_seed
        super().__init__(max_nodes=max_nodes, gamma=gamma, projection=True, **kwargs)

    def _build(self, n_games):
        super()._build(n_games)
        if self.random_seed is None and self.n_games == 1:
            states = [random.getstate()]
        else:
            base = 0 if self.random_seed is None else int(self.random_seed)
            states = [random.Random(base + g).getstate() for g in range(self.n_games)]
        self.store.set_python_random_states(states)

    def search_model(self):
        return None   # rollouts happen inside the tree kernel: the native loop launches no evaluator

    def evaluate_requests(self):
        pass
