#!/bin/bash
# timing of build variants (variants/libtetris_<name>.so), same box, default bench window
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = base ]; then unset TETRIS_MCTS_LIB; else export TETRIS_MCTS_LIB=$PWD/variants/libtetris_$v.so; fi
  timeout 200 python bench.py --no-cpu-baseline > $OUT/v.$v.json 2> $OUT/v.$v.err
  python - <<PY
import json
d=json.load(open("$OUT/v.$v.json"))
k=d["last_sim_phase_kcycles"]
print("$v", round(d["ms_per_step"],2), "tree", round(d["roofline_other"]["avg_launch_ms"],4), "nn", round(d["roofline"]["avg_launch_ms"],4), "sel/back/exp", round(k["CYC_SELECT"],1), round(k["CYC_BACK"],1), round(k["CYC_EXPAND"],1), "err", d["error_games"])
PY
done
