"""The reference's module name for its distributional agent (agents/DistValueSimOnline.py defines class DistValueSim; play.py
would look for a class named like the module): both names resolve to the MI355X engine's DistValueSim."""
from tetris_mcts_amd.agents.DistValueSim import DistValueSim  # noqa: F401
DistValueSimOnline = DistValueSim
