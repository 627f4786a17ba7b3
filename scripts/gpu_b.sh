#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --maxfail=12 --durations=12 > $OUT/b.pytest.log 2>&1
echo "pytest rc=$?"; tail -n 30 $OUT/b.pytest.log | cut -c1-300
for g in 4096 1024; do
  TETRIS_MCTS_LIB=$PWD/variants/libtetris_prof.so timeout 600 python bench.py --games $g --no-cpu-baseline --steps 10 > $OUT/b.prof_g$g.json 2> $OUT/b.prof_g$g.err
  echo "prof g=$g rc=$?"; python - <<PY
import json
d=json.load(open("$OUT/b.prof_g$g.json"))
print({k:d[k] for k in ("value","ms_per_step","mean_trace_len","max_trace_len","last_sim_phase_kcycles")})
print("tree ms", d["roofline"]["avg_launch_ms"] if d["roofline"]["bound"]=="hbm" else d["roofline_other"]["avg_launch_ms"], "nn ms", d["roofline"]["avg_launch_ms"] if d["roofline"]["bound"]=="mfma" else d["roofline_other"]["avg_launch_ms"])
PY
done
