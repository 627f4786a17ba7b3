#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=6 -k "${PYTEST_K:-sampled_seeds or valuenet or value_net}" > $OUT/k.pytest.log 2>&1
echo "pytest rc=$?"; tail -n 6 $OUT/k.pytest.log | cut -c1-300
for sp in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --split $sp > $OUT/k.bench_$sp.json 2> $OUT/k.bench_$sp.err
python - <<PY
import json
d=json.load(open("$OUT/k.bench_$sp.json"))
print($sp, {k:d.get(k) for k in ("value","ms_per_step","mean_trace_len")})
for r in ("roofline","roofline_other"): print("   ", d[r]["kernel"][:40], d[r]["avg_launch_ms"], round(d[r]["frac"],4))
PY
done
