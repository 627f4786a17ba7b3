"""ORACLE — test infrastructure only.

CPU restatements of the reference hot path (engine spec, UCT kernels, tree agents, value net)
plus the build recipe for the reference's own native sources (oracle/_ref/).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; nothing under
tetris_mcts_amd/ does.
"""
