#!/bin/bash
# Validate a build variant on the GPU box: scripts/gpu_vt.sh NAME [pytest -k expression]
#   (variants/libtetris_NAME.so from scripts/build_variant.sh NAME -D...): the GPU suite (5 xdist workers) against that
#   library, then the default bench window.  E.g. first thing next round:
#     scripts/build_variant.sh walk2 -DTM_WALK2;   gpurun --timeout 600 -- 'bash scripts/gpu_vt.sh walk2'
#     scripts/build_variant.sh overlap -DTM_OVERLAP; gpurun --timeout 300 -- 'bash scripts/gpu_vt.sh overlap "sampled_seeds or with_gc"'
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
V=$1; K=${2:-}
export TETRIS_MCTS_LIB=$PWD/variants/libtetris_$V.so
if [ -n "$K" ]; then
  timeout 900 python -m pytest tests -m gpu -q -n 5 --maxfail=8 -k "$K" > $OUT/vt.$V.pytest.log 2>&1
else
  timeout 900 python -m pytest tests -m gpu -q -n 5 --maxfail=8 > $OUT/vt.$V.pytest.log 2>&1
fi
echo "pytest rc=$?"; tail -n 8 $OUT/vt.$V.pytest.log | cut -c1-250
timeout 200 python bench.py --no-cpu-baseline > $OUT/vt.$V.bench.json 2> $OUT/vt.$V.bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$OUT/vt.$V.bench.json")); k=d["last_sim_phase_kcycles"]
print("$V", round(d["ms_per_step"],2), "tree", round(d["roofline_other"]["avg_launch_ms"],4), "nn", round(d["roofline"]["avg_launch_ms"],4), "err", d["error_games"], "len", d["mean_trace_len"], d["max_trace_len"], "exp/s", round(d["value"]))
PY
