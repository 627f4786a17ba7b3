#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
TAG=${1:-rv}
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/$TAG.smoke.log 2>&1; echo "smoke rc=$?"; tail -n 1 $OUT/$TAG.smoke.log | cut -c1-300
timeout 900 python -m pytest tests -m gpu -q -n 6 --maxfail=6 -k "valuenet or value_net_in_the_loop or sampled_seeds or real_4096 or with_gc" > $OUT/$TAG.pytest.log 2>&1
echo "pytest rc=$?"; tail -n 8 $OUT/$TAG.pytest.log | cut -c1-300
bash scripts/gpu_r03_g.sh $TAG
