#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=6 -k "${PYTEST_K:-sampled_seeds or traces_beyond or hash_eval_with_gc or smoke}" > $OUT/d.pytest.log 2>&1
echo "pytest rc=$?"; tail -n 15 $OUT/d.pytest.log | cut -c1-300
timeout 900 python bench.py --no-cpu-baseline > $OUT/d.bench.json 2> $OUT/d.bench.err
echo "bench rc=$?"; python - <<PY
import json
d=json.load(open("$OUT/d.bench.json"))
print({k:d.get(k) for k in ("value","ms_per_step","mean_trace_len","max_trace_len","last_sim_phase_kcycles","walk_mispredicted_levels")})
for r in ("roofline","roofline_other"): print(d[r]["kernel"][:40], d[r]["avg_launch_ms"], d[r]["frac"])
PY
TETRIS_MCTS_LIB=$PWD/variants/libtetris_prof.so timeout 600 python bench.py --no-cpu-baseline --steps 10 > $OUT/d.prof.json 2> $OUT/d.prof.err
python - <<PY
import json
d=json.load(open("$OUT/d.prof.json"))
print("prof", {k:d.get(k) for k in ("value","ms_per_step","last_sim_phase_kcycles","walk_mispredicted_levels")})
PY
