#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=6 -k "${PYTEST_K:-sampled_seeds or hash_eval_with_gc}" > $OUT/i.pytest.log 2>&1
echo "pytest rc=$?"; tail -n 6 $OUT/i.pytest.log | cut -c1-300
for g in 4096 1024; do
timeout 600 python bench.py --no-cpu-baseline --games $g --steps ${STEPS:-20} > $OUT/i.bench_$g.json 2> $OUT/i.bench_$g.err
python - <<PY
import json
d=json.load(open("$OUT/i.bench_$g.json"))
print($g, {k:d.get(k) for k in ("value","ms_per_step","mean_trace_len","last_sim_phase_kcycles","walk_mispredicted_levels")})
for r in ("roofline","roofline_other"): print("   ", d[r]["kernel"][:40], d[r]["avg_launch_ms"])
PY
done
