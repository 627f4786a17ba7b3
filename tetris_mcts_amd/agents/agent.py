"""Batched mirror of the reference's `Agent` / `TreeAgent` (agents/agent.py:10-307).

Same surface — play(), get_action(), get_prob(), get_stats(), get_value_and_variance(), update_root(game),
close() — over `n_games` concurrent games whose trees live on the GPU (store.TreeStore).  With n_games == 1
return values have the reference's scalar shapes, so play.py drives it unchanged.
"""
import numpy as np
import torch

from .. import store as st


class Agent:
    def __init__(self, n_actions=7, benchmark=False, **kwargs):
        self.episode = 0
        self.n_actions = n_actions
        self.benchmark = benchmark

    def play(self):
        raise NotImplementedError('update_root not implemented')

    def get_action(self):
        raise NotImplementedError('get_action not implemented')

    def get_prob(self):
        raise NotImplementedError('get_action not implemented')

    def update_root(self, game):
        raise NotImplementedError('update_root not implemented')

    def close(self):
        raise NotImplementedError('close not implemented')


class TreeAgent(Agent):
    kind = st.KIND_VALUESIM
    low = 1

    def __init__(self, sims=100, max_nodes=500000, env=None, env_args=None, node_saver=None, projection=True,
                 min_visits=30, n_games=None, gamma=0.999, online=False, min_visits_to_store=10, replay_cap=0,
                 max_trace=1024, nq_size=1 << 20, reset_on_pool_exhaustion=True, n_sub=1, ev_every=0,
                 gc_slice_cycles=150000, gc_spec_nodes=None, gc_cost_units=0, gc_collectors=0, **kwargs):
        super().__init__(**kwargs)
        if not projection:
            raise NotImplementedError("projection=False is broken in the reference itself (ValueSim.py:73-74)")
        self.sims = sims
        self.max_nodes = max_nodes
        self.env, self.env_args = env, env_args or ((20, 10), 1, 0, 0)
        self.min_visits = min_visits
        self.node_saver = node_saver
        self.projection = True
        self.gamma = gamma
        self.n_games = n_games
        self._store_kwargs = dict(kind=self.kind, env_args=self.env_args, gamma=gamma, low=self.low, online=online,
                                  min_visits_to_store=min_visits_to_store, replay_cap=replay_cap, max_trace=max_trace,
                                  nq_size=nq_size, gc_slice_cycles=gc_slice_cycles, gc_spec_nodes=gc_spec_nodes,
                                  gc_cost_units=gc_cost_units, gc_collectors=gc_collectors)
        self.n_sub, self.ev_every = int(n_sub), int(ev_every)
        self._pending_events, self.loop_events, self.catchup_launches = [], dict(timed=0, nn_ms_sum=0.0, tree_ms_sum=0.0), 0
        self.store = None
        self.reset_on_pool_exhaustion = reset_on_pool_exhaustion
        self._pending_pool_reset = None
        self.stats = None
        if n_games is not None:
            self._build(n_games)

    def _build(self, n_games):
        self.n_games = int(n_games)
        self.n_sub = max(1, min(self.n_sub, self.n_games // 4))     # a sub-batch is at least one workgroup of the tree kernel
        self.store = st.TreeStore(self.n_games, self.max_nodes, **self._store_kwargs)

    # ---- evaluation hook (the `evaluator` callable of MCTSAgent, agent.cpp:396-405) ----
    def evaluate(self, states, v_out, var_out):
        raise NotImplementedError('evaluate not implemented for this agent.')

    def evaluate_requests(self):
        """Evaluate the store's pending leaf requests into eval_v / eval_var.  Default: render the observations
        to int8 [B,200] and call evaluate(); agents with the HIP value net override this with the fused path."""
        s = self.store
        states = s.render_eval()
        self.evaluate(states, s.t["eval_v"], s.t["eval_var"])

    def search_model(self):
        """The model whose HIP value net the native launch loop (search.hip) evaluates leaves with; False = this agent's
        evaluator is a Python callable, so the loop runs here instead."""
        return False

    def mcts(self, sims):
        """`sims` simulations for every game (TreeAgent.play -> self.mcts(self.root, self.sims), agents/agent.py:147-150).
        A launch starts a simulation only for games with quota left; a game that collects garbage loses launches and
        catches up at the end (store.sims_remaining)."""
        s = self.store
        model = self.search_model()
        if model is not False:
            s.search(sims, model, n_sub=self.n_sub, ev_every=self.ev_every)
            return
        both = st.SIM_BACKUP | st.SIM_FRONT
        s.move_begin(sims)
        s.sim_step(both)
        todo, first = sims, True
        ev = self.ev_every if first else 0
        while todo > 0:
            for i in range(todo):
                if first and ev and i % ev == 0:
                    # events around the evaluator and the tree kernel of every ev-th simulation (the Python-driven loop's
                    # counterpart of tm_search_stats; summed in self.loop_events and read by the benchmark)
                    import torch
                    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                    e[0].record()
                    self.evaluate_requests()
                    e[1].record()
                    s.sim_step(both)
                    e[2].record()
                    self._pending_events.append(e)
                    if len(self._pending_events) >= 256:
                        self._fold_events()                 # bounded whether or not anybody reads loop_stats()
                else:
                    self.evaluate_requests()
                    s.sim_step(both)
            if not first:
                self.catchup_launches += todo
            first = False
            todo, collecting = s.sims_remaining()
            while collecting:                      # catch-up: collections under way are finished by collector-only launches
                for _ in range(6):
                    s.gc_step()
                todo, collecting = s.sims_remaining()

    def _fold_events(self, wait=False):
        """completed event triples into the running sums (all of them after a synchronize)"""
        import torch
        if wait:
            torch.cuda.synchronize()
        keep = []
        for e in self._pending_events:
            if wait or e[2].query():
                self.loop_events["nn_ms_sum"] += e[0].elapsed_time(e[1])
                self.loop_events["tree_ms_sum"] += e[1].elapsed_time(e[2])
                self.loop_events["timed"] += 1
            else:
                keep.append(e)
        self._pending_events = keep

    def loop_stats(self, reset=True):
        """Python-driven loop only: dict(timed, nn_ms_sum, tree_ms_sum, catchup_launches) from the sampled event triples"""
        self._fold_events(wait=True)
        out = dict(self.loop_events, catchup_launches=self.catchup_launches)
        if reset:
            self.loop_events = dict(timed=0, nn_ms_sum=0.0, tree_ms_sum=0.0)
            self.catchup_launches = 0
        return out

    def play(self):
        self.mcts(self.sims)
        return self.get_action()

    def compute_stats(self):
        stats, action = self.store.root_stats()
        return stats, action

    def get_action(self):
        stats, action = self.compute_stats()
        self._stats_dev, self._action_dev = stats, action
        self.stats = None
        err = self.store.errors()
        a = action.cpu().numpy()
        if bool((err != 0).any().item()):
            pool = (err & 1) != 0
            if self.reset_on_pool_exhaustion and bool(((err & ~1) == 0).all().item()):
                # The reachable tree outgrew the pool (the reference prints "MAX_NODES EXCEEDED" and runs into undefined
                # behaviour, agent.cpp:227-231).  Those games keep the action of the search done so far and restart
                # with an empty tree at their next update_root.
                from sys import stderr
                print("MAX_NODES EXCEEDED in %d game(s): their trees are re-initialised" % int(pool.sum().item()),
                      file=stderr, flush=True)
                self._pending_pool_reset = pool.clone()
            else:
                raise RuntimeError("tree engine error flags: %s" % sorted(set(err.cpu().numpy().tolist())))
        return int(a[0]) if self.n_games == 1 else a

    def get_stats(self):
        if self.stats is None:
            self.stats = self._stats_dev.cpu().numpy()
        return np.copy(self.stats[0]) if self.n_games == 1 else np.copy(self.stats)

    def get_prob(self):
        s = self.get_stats().reshape(-1, 3, 7)
        p = s[:, 0] / s[:, 0].sum(axis=1, keepdims=True)
        return p[0] if self.n_games == 1 else p

    def new_node(self, game):
        """Node index of `game`'s state in every tree, inserted if new (agents/agent.py:90-130, agent.cpp:265-270);
        an int with one game, else an int32 array [n_games]."""
        if self.store is None:
            self._build(getattr(game, "n_games", 1))
        idx = self.store.new_node(game.games).cpu().numpy()
        return int(idx[0]) if self.n_games == 1 else idx

    def expand(self, game):
        """idx = new_node(game); child[idx][a] = new_node(game after action a) (agents/agent.py:136-145, agent.cpp:201-218)."""
        if self.store is None:
            self._build(getattr(game, "n_games", 1))
        self.store.new_node(game.games, expand=True)

    def remove_nodes(self):
        """Collect everything unreachable from the root now (agents/agent.py:246-257); harvests replay tuples when online."""
        self.store.remove_nodes()

    def get_value_and_variance(self, node=None):
        """(value, variance) of a node's observation (agents/agent.py:195-204); node=None is the root, else a node index
        (one game) or an array of n_games indices."""
        s = self.store
        g = torch.arange(self.n_games, device=s.device)
        if node is None:
            root = s.t["gs"][:, st.GS["ROOT"]].long()
        else:
            root = torch.as_tensor(np.atleast_1d(node), device=s.device).long().reshape(self.n_games)
        o = s.t["node_rec"][g, root, 29].long()
        vv = s.t["obs_stat"][g, o, 1:3].view(torch.float32).cpu().numpy()
        return (vv[0, 0], vv[0, 1]) if self.n_games == 1 else (vv[:, 0], vv[:, 1])

    def update_root(self, game):
        if self.store is None:
            self._build(getattr(game, "n_games", 1))
        self.store.set_root_games(game.games)
        if self._pending_pool_reset is not None:
            self.store.pool_reset(self._pending_pool_reset)
            self._pending_pool_reset = None
        self.store.update_root()
        ended = np.atleast_1d(game.end)
        self.episode += int(ended.sum())

    def close(self):
        pass
