#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=6 -k "traces_beyond or hash_eval_with_gc or replay_harvest" > $OUT/l.pytest.log 2>&1
echo "pytest rc=$?"; tail -n 5 $OUT/l.pytest.log | cut -c1-300
for v in prio noprio; do
if [ $v = prio ]; then LIB=""; else LIB=$PWD/variants/libtetris_noprio.so; fi
TETRIS_MCTS_LIB=$LIB timeout 600 python bench.py --no-cpu-baseline > $OUT/l.bench_$v.json 2> $OUT/l.bench_$v.err
python - <<PY
import json
d=json.load(open("$OUT/l.bench_$v.json"))
print("$v", {k:d.get(k) for k in ("value","ms_per_step","mean_trace_len")})
for r in ("roofline","roofline_other"): print("   ", d[r]["kernel"][:40], d[r]["avg_launch_ms"], round(d[r]["frac"],4))
PY
done
# steady state: GCs inside the window
timeout 900 python bench.py --no-cpu-baseline --warmup 40 --steps 20 > $OUT/l.bench_steady.json 2> $OUT/l.bench_steady.err
python - <<PY
import json
d=json.load(open("$OUT/l.bench_steady.json"))
print("steady", {k:d.get(k) for k in ("value","ms_per_step","mean_trace_len","max_trace_len","gc","episodes_finished","lines_cleared_per_episode")})
for r in ("roofline","roofline_other"): print("   ", d[r]["kernel"][:40], d[r]["avg_launch_ms"], round(d[r]["frac"],4))
PY
