#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python bench.py --no-cpu-baseline > $OUT/m.bench_prio.json 2> $OUT/m.bench_prio.err
python - <<PY
import json
d=json.load(open("$OUT/m.bench_prio.json"))
print("prio", {k:d.get(k) for k in ("value","ms_per_step","mean_trace_len")})
for r in ("roofline","roofline_other"): print("   ", d[r]["kernel"][:40], d[r]["avg_launch_ms"], round(d[r]["frac"],4))
PY
timeout 900 python bench.py --no-cpu-baseline --warmup 75 --steps 20 > $OUT/m.bench_steady.json 2> $OUT/m.bench_steady.err
python - <<PY
import json
d=json.load(open("$OUT/m.bench_steady.json"))
print("steady", {k:d.get(k) for k in ("value","ms_per_step","mean_trace_len","max_trace_len","gc","episodes_finished","lines_cleared_per_episode")})
for r in ("roofline","roofline_other"): print("   ", d[r]["kernel"][:40], d[r]["avg_launch_ms"], round(d[r]["frac"],4))
PY
