#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
TAG=${1:-sd}
timeout 900 python -m pytest tests -m gpu -q -n 4 --maxfail=6 -k "many_games or dist_agent or online_training_loop or with_gc or beyond_128 or real_4096 or pool" > $OUT/$TAG.pytest.log 2>&1
echo "pytest rc=$?"; tail -n 8 $OUT/$TAG.pytest.log | cut -c1-300
timeout 300 python bench.py --no-cpu-baseline --warmup 75 --steps 20 --steady-steps 0 > $OUT/$TAG.steady.json 2> $OUT/$TAG.steady.err
python - <<PY
import json
d=json.load(open("$OUT/$TAG.steady.json")); g=d["gc"]
print("steady ms/move", round(d["ms_per_step"],1), "tree", round(d["roofline_other"]["avg_launch_ms"],4), "trace", d["mean_trace_len"], g)
PY
timeout 300 python scripts/gc_lag.py --moves 3 > $OUT/$TAG.lag.log 2>&1; grep -E "^move|one collection|catch-up" $OUT/$TAG.lag.log
