#!/bin/bash
# the whole GPU suite + smoke: the small tests on six xdist workers (the suite's time is mostly the CPU oracle's), the tests
# that hold a 4096-game store (130 - 230 GiB each) one after the other
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
TAG=${1:-full}
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/$TAG.smoke.log 2>&1; echo "smoke rc=$?"; tail -n 1 $OUT/$TAG.smoke.log | cut -c1-300
timeout 1500 python -m pytest tests -m gpu -q -n 6 --maxfail=12 --durations=6 -k "not (4096 or benchmarked_dist)" > $OUT/$TAG.pytest.log 2>&1
echo "pytest rc=$?"; tail -n 14 $OUT/$TAG.pytest.log | cut -c1-300
timeout 1500 python -m pytest tests -m gpu -q --durations=6 -k "4096 or benchmarked_dist" > $OUT/$TAG.pytest_big.log 2>&1
echo "pytest (4096-game stores) rc=$?"; tail -n 10 $OUT/$TAG.pytest_big.log | cut -c1-300
