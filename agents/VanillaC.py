"""`agents.VanillaC` as the reference's play.py resolves it (`import_module('agents.' + name)`, play.py:81-82): the MI355X
engine's class under the reference's module path.  `agents/` has no __init__.py on purpose (a namespace package, like the
reference's), so it never shadows another `agents` directory that comes first on sys.path."""
from tetris_mcts_amd.agents.VanillaC import VanillaC  # noqa: F401
