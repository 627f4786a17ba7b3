"""tetris_mcts_amd — MI355X-native batched Tetris-MCTS self-play engine.

Drop-in for the hot path of hrpan/tetris_mcts: the Tetris board step and the UCT
select / expand / backup loop run as hand-written HIP kernels (one wavefront per game), leaf states are
batched into the value network, games shard across GPUs.  Host-side classes mirror the reference's:
`pyTetris.Tetris`, `agents.ValueSim.ValueSim`, `agents.ValueSimLP.ValueSimLP`, `model.Model_VV`.
"""
__version__ = "0.1.0"
