#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python __graft_entry__.py smoke > $OUT/full.smoke.log 2>&1; echo "smoke rc=$?"
timeout 2400 python -m pytest tests -m gpu -q --maxfail=12 --durations=8 > $OUT/full.pytest.log 2>&1
echo "pytest rc=$?"; tail -n 16 $OUT/full.pytest.log | cut -c1-300
