#!/bin/bash
# steady-state sweep of the collectors' time allowance
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for sl in ${SLICES:-200000 300000 500000}; do
  timeout 300 python bench.py --no-cpu-baseline --warmup 75 --steps 20 --gc-slice-cycles $sl > $OUT/rn.steady_$sl.json 2> $OUT/rn.steady_$sl.err
  python - <<PY
import json
d=json.load(open("$OUT/rn.steady_$sl.json"))
print("slice $sl ms/move", round(d["ms_per_step"],1), "tree", round(d["roofline_other"]["avg_launch_ms"],4), d["gc"])
PY
done
