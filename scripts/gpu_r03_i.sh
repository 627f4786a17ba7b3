#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
TAG=${1:-rw}
timeout 900 python -m pytest tests/test_gpu_dist_agent.py -m gpu -q -x > $OUT/$TAG.dist.log 2>&1; echo "dist rc=$?"; tail -n 30 $OUT/$TAG.dist.log | cut -c1-400
bash scripts/gpu_r03_h.sh $TAG
