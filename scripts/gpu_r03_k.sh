#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
TAG=${1:-sb}
timeout 900 python -m pytest tests -m gpu -q -n 4 --maxfail=6 -k "many_games or dist_agent or online_training_loop or with_gc or beyond_128 or real_4096" > $OUT/$TAG.pytest.log 2>&1
echo "pytest rc=$?"; tail -n 8 $OUT/$TAG.pytest.log | cut -c1-300
for spec in default 256 0; do
  arg=""; [ $spec != default ] && arg="--gc-spec-nodes $spec"
  timeout 300 python bench.py --no-cpu-baseline --warmup 75 --steps 20 --steady-steps 0 $arg > $OUT/$TAG.steady_$spec.json 2> $OUT/$TAG.steady_$spec.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/$TAG.steady_$spec.json"))
    print("spec=$spec steady ms/move", round(d["ms_per_step"],1), "tree", round(d["roofline_other"]["avg_launch_ms"],4), "trace", d["mean_trace_len"], d["gc"])
except Exception as e:
    print("spec=$spec failed", e)
PY
done
