"""`agents.DistValueSim`: the distributional agent rebuilt from the working parts of the reference's
agents/DistValueSimOnline.py (whose class is named DistValueSim and which does not import)."""
from tetris_mcts_amd.agents.DistValueSim import DistValueSim  # noqa: F401
