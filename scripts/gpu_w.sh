#!/bin/bash
# one shot at the experimental walk (variants/libtetris_walk2.so): parity subset + bench
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
export TETRIS_MCTS_LIB=$PWD/variants/libtetris_walk2.so
timeout 50 python -m pytest tests -m gpu -q -n 5 --maxfail=3 -k "(sampled_seeds and ValueSim-20) or with_gc or reference_golden_runs or vanilla_batch" > $OUT/w.pytest.log 2>&1
echo "pytest rc=$?"; tail -n 6 $OUT/w.pytest.log | cut -c1-300
timeout 40 python bench.py --no-cpu-baseline > $OUT/w.bench.json 2> $OUT/w.bench.err
python - <<PY
import json
d=json.load(open("$OUT/w.bench.json")); k=d["last_sim_phase_kcycles"]
print("walk2", round(d["ms_per_step"],2), "tree", round(d["roofline_other"]["avg_launch_ms"],4), "nn", round(d["roofline"]["avg_launch_ms"],4), "sel/back/exp", round(k["CYC_SELECT"],1), round(k["CYC_BACK"],1), round(k["CYC_EXPAND"],1), "err", d["error_games"], "miss", round(d["walk_mispredicted_levels"],4), "len", round(d["mean_trace_len"],2), d["max_trace_len"], "exp/s", round(d["value"]))
PY
