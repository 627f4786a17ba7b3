#!/bin/bash
# the whole GPU suite (xdist workers share the GPU; the suite's time is mostly the CPU oracle's) + smoke
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
TAG=${1:-full}
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/$TAG.smoke.log 2>&1; echo "smoke rc=$?"; tail -n 1 $OUT/$TAG.smoke.log | cut -c1-300
timeout 1500 python -m pytest tests -m gpu -q -n 6 --maxfail=12 --durations=6 > $OUT/$TAG.pytest.log 2>&1
echo "pytest rc=$?"; tail -n 14 $OUT/$TAG.pytest.log | cut -c1-300
