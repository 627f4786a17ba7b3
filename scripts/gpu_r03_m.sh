#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
TAG=${1:-sf}
timeout 300 python bench.py --no-cpu-baseline --steady-steps 0 > $OUT/$TAG.fresh.json 2> $OUT/$TAG.fresh.err
timeout 300 python bench.py --no-cpu-baseline --warmup 55 --steps 15 --steady-steps 0 > $OUT/$TAG.pre.json 2> $OUT/$TAG.pre.err
python - <<PY
import json
for n in ("fresh","pre"):
    d=json.load(open("$OUT/$TAG.%s.json"%n)); g=d["gc"]
    print(n, "ms/move", round(d["ms_per_step"],2), "value", round(d["value"]/1e6,3), "tree", round(d["roofline_other"]["avg_launch_ms"],4), "nn", round(d["roofline"].get("avg_launch_ms",0),4), "trace", round(d["mean_trace_len"],2), "prefix", round(d["walk_levels_taken_over_from_the_previous_walk"],3), "collections", g["collections"], "catchup", g["catchup_launches_per_move"])
PY
