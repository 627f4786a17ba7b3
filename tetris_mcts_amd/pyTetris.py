"""`Tetris` — the environment class the reference imports from the external pyTetris package
(play.py:1,75-76; agents/agent.py:70), here a batch of `n_games` device-resident games stepped by the HIP
engine (ENGINE_SPEC.md).  With n_games == 1 the attributes are scalars exactly as in pyTetris, so
play.py's loop runs unchanged; with n_games > 1 they are numpy arrays of length n_games.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .store import _p, _stream


class Tetris:
    def __init__(self, shape=(20, 10), actions_per_drop=1, scoring=0, randomizer=0, seed=0, n_games=1, device="cuda"):
        if tuple(shape) != (20, 10):
            raise ValueError("shape must be (20, 10)")
        if not torch.cuda.is_available():
            raise RuntimeError("tetris_mcts_amd needs a ROCm GPU (gfx950); there is no CPU path")
        self.L = _lib.lib()
        self.n_games = int(n_games)
        self.env_args = ((20, 10), int(actions_per_drop), int(scoring), int(randomizer))
        self.device = torch.device(device)
        self.games = torch.zeros(self.n_games, 16, dtype=torch.int32, device=self.device)
        self._ls = torch.zeros(self.n_games, 4, dtype=torch.int32, device=self.device)
        self._info = torch.zeros(self.n_games, 8, dtype=torch.int32, device=self.device)
        self._state = torch.zeros(self.n_games, 200, dtype=torch.int8, device=self.device)
        self._actions = torch.zeros(self.n_games, dtype=torch.int32, device=self.device)
        s = _lib.TmStore()
        s.n_games, s.app, s.scoring, s.randomizer = self.n_games, self.env_args[1], self.env_args[2], self.env_args[3]
        s.env_game, s.env_line_stats = self.games.data_ptr(), self._ls.data_ptr()
        self.s = s
        self._host = None
        self.seed(seed)

    # ---- pyTetris surface ----
    def seed(self, seed):
        """seed: int (game g gets seed+g) or an array of n_games uint32 seeds."""
        if np.isscalar(seed):
            seeds = (int(seed) + np.arange(self.n_games, dtype=np.int64)) & 0xFFFFFFFF
        else:
            seeds = np.asarray(seed, dtype=np.int64) & 0xFFFFFFFF
        t = torch.from_numpy(seeds.astype(np.uint32).view(np.int32)).to(self.device)
        _lib.check(self.L.tm_env_init(C.byref(self.s), _p(t), _stream()), "tm_env_init")
        self._host = None

    def play(self, action):
        if torch.is_tensor(action):
            self._actions.copy_(action.to(torch.int32).reshape(-1))
        else:
            a = np.asarray(action, dtype=np.int32).reshape(-1)
            if a.size == 1 and self.n_games > 1:
                a = np.full(self.n_games, int(a[0]), np.int32)
            self._actions.copy_(torch.from_numpy(a))
        _lib.check(self.L.tm_env_step(C.byref(self.s), _p(self._actions), _stream()), "tm_env_step")
        self._host = None

    def reset(self, mask=None):
        """reset() restarts every game (pyTetris); reset(mask) only where mask is true; reset('ended') the ended ones."""
        if mask is None:
            m = torch.ones(self.n_games, dtype=torch.uint8, device=self.device)
            ptr = _p(m)
        elif isinstance(mask, str):
            ptr = C.c_void_p(0)
        else:
            m = torch.as_tensor(mask, device=self.device).to(torch.uint8)
            ptr = _p(m)
        _lib.check(self.L.tm_env_reset(C.byref(self.s), ptr, _stream()), "tm_env_reset")
        self._host = None

    def copy_from(self, other):
        self.games.copy_(other.games)
        self._ls.copy_(other._ls)
        self._host = None

    def getState(self):
        _lib.check(self.L.tm_env_render(C.byref(self.s), _p(self._state), _stream()), "tm_env_render")
        a = self._state.cpu().numpy().reshape(self.n_games, 20, 10)
        return a[0] if self.n_games == 1 else a

    def printState(self):
        s = self.getState().reshape(-1, 20, 10)[0]
        print("\n".join("".join(".#@"[v] for v in row) for row in s), flush=True)

    def _sync(self):
        if self._host is None:
            _lib.check(self.L.tm_env_info(C.byref(self.s), _p(self._info), _stream()), "tm_env_info")
            self._host = self._info.cpu().numpy()
        return self._host

    def _field(self, col, cast):
        h = self._sync()
        return cast(h[0, col]) if self.n_games == 1 else h[:, col].astype(cast if cast is not bool else np.bool_)

    end = property(lambda self: self._field(0, bool))
    score = property(lambda self: self._field(1, int))
    line_clears = property(lambda self: self._field(2, int))
    combo = property(lambda self: self._field(3, int))

    @property
    def line_stats(self):
        h = self._sync()
        return h[0, 4:8].copy() if self.n_games == 1 else h[:, 4:8].copy()

    def packed(self):
        return self.games.cpu().numpy().view(np.uint32)

    def __eq__(self, other):
        return isinstance(other, Tetris) and bool(torch.equal(self.games, other.games))

    def __hash__(self):
        return hash(self.games.cpu().numpy().tobytes())
