"""`from pyTetris import Tetris` (play.py:1, agents/agent.py:70) resolved to the MI355X engine's batched environment.
The CPU oracle has a compiled extension of the same name (oracle/_build, test infrastructure): checker code loads that
one by path (oracle.binding.oracle_pytetris()) and never through this name."""
from tetris_mcts_amd.pyTetris import Tetris  # noqa: F401
