#!/bin/bash
# r03 quick check: smoke, tree parity subset (or "$1" as the -k expression), default bench, walk-prefix statistics
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
TAG=${2:-rb}
K=${1:-"sampled_seeds or with_gc or app2 or app3 or value_net_in_the_loop or vanilla_batch or golden_runs or beyond_128"}
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/$TAG.smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $OUT/$TAG.smoke.log | cut -c1-400
timeout 900 python -m pytest tests -m gpu -q -n 6 --maxfail=6 -k "$K" > $OUT/$TAG.pytest.log 2>&1
echo "pytest rc=$?"; tail -n 12 $OUT/$TAG.pytest.log | cut -c1-300
timeout 200 python bench.py --no-cpu-baseline > $OUT/$TAG.bench.json 2> $OUT/$TAG.bench.err; echo "bench rc=$?"
python - <<PY
import json
try:
    d=json.load(open("$OUT/$TAG.bench.json"))
    print("ms/move", round(d["ms_per_step"],2), "tree", round(d["roofline_other"]["avg_launch_ms"],4), "nn", round(d["roofline"]["avg_launch_ms"],4), "err", d["error_games"], "len", d["mean_trace_len"], d["max_trace_len"], "exp/s", round(d["value"]), d["last_sim_phase_kcycles"])
except Exception as e: print("bench failed", e)
PY
tail -n 3 $OUT/$TAG.bench.err | cut -c1-300
