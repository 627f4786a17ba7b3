"""`from pyTetris import Tetris` (play.py:1, agents/agent.py:70) resolved to the MI355X engine's batched environment.
The CPU oracle's own `pyTetris` extension (oracle/_build, test infrastructure) is only ever imported from a path that is
put in front of this one."""
from tetris_mcts_amd.pyTetris import Tetris  # noqa: F401
