#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for g in 1024 2048 8192; do
timeout 600 python bench.py --no-cpu-baseline --games $g --steps 10 > $OUT/g.bench_$g.json 2> $OUT/g.bench_$g.err
python - <<PY
import json
d=json.load(open("$OUT/g.bench_$g.json"))
print($g, {k:d.get(k) for k in ("value","ms_per_step","mean_trace_len","last_sim_phase_kcycles")})
for r in ("roofline","roofline_other"): print("   ", d[r]["kernel"][:40], d[r]["avg_launch_ms"])
PY
done
