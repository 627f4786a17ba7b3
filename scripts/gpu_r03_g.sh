#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
TAG=${1:-rs}
timeout 300 python bench.py --no-cpu-baseline > $OUT/$TAG.bench.json 2> $OUT/$TAG.bench.err; echo "rc=$?"
python - <<PY
import json
d=json.load(open("$OUT/$TAG.bench.json"))
print("fresh ms/move", round(d["ms_per_step"],2), "tree", round(d["roofline_other"]["avg_launch_ms"],4), "nn", round(d["roofline"]["avg_launch_ms"],4), "err", d["error_games"])
s=d["steady_state"]; print("steady ms/move", round(s["ms_per_step"],2), "tree", round(s["tree_kernel_ms"],4), "nn", round(s["value_net_ms"],4), "err", s["error_games"], {k:(round(v,1) if isinstance(v,float) else v) for k,v in s["gc"].items()})
PY
