#!/bin/bash
# last check of the round on the final tree: whole GPU suite, smoke, and the online (harvest + exchange) leg in the full-pool regime
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python __graft_entry__.py smoke > $OUT/j.smoke.log 2>&1; echo "smoke rc=$?"
timeout 600 python -m pytest tests -m gpu -q -n 5 --maxfail=12 --durations=5 > $OUT/j.pytest.log 2>&1
echo "pytest rc=$?"; tail -n 10 $OUT/j.pytest.log | cut -c1-250
( time timeout 300 python bench.py --no-cpu-baseline --online --warmup 75 --steps 20 > $OUT/j.bench_online.json 2> $OUT/j.bench_online.err ) 2>&1 | grep real
python -c "
import json; d=json.load(open('$OUT/j.bench_online.json')); print({k:d.get(k) for k in ('value','ms_per_step','gc','exchange')})"
