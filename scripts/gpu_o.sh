#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=6 -k "sampled_seeds or app2 or app3 or cpp_agent_golden" > $OUT/o.pytest.log 2>&1
echo "pytest rc=$?"; tail -n 5 $OUT/o.pytest.log | cut -c1-300
timeout 600 python bench.py --no-cpu-baseline > $OUT/o.bench.json 2> $OUT/o.bench.err
python - <<PY
import json
d=json.load(open("$OUT/o.bench.json"))
print({k:d.get(k) for k in ("value","ms_per_step","last_sim_phase_kcycles")})
for r in ("roofline","roofline_other"): print("   ", d[r]["kernel"][:40], d[r]["avg_launch_ms"], round(d[r]["frac"],4))
PY
