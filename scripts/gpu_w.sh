#!/bin/bash
# the pinned-register two-level walk (variants/libtetris_walk2.so): benched-regime parity test
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
export TETRIS_MCTS_LIB=$PWD/variants/libtetris_walk2.so
timeout 19 python -m pytest tests/test_gpu_benched_regime.py -m gpu -q -x -k "sampled_seeds and ValueSim-20" > $OUT/w2.pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 $OUT/w2.pytest.log | cut -c1-200
