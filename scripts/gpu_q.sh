#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -n 4 --maxfail=6 -k "valuenet or value_net or (sampled_seeds and ValueSim-20)" > $OUT/q.pytest.log 2>&1
echo "pytest rc=$?"; tail -n 30 $OUT/q.pytest.log | cut -c1-400
timeout 300 python bench.py --no-cpu-baseline > $OUT/q.bench.json 2> $OUT/q.bench.err
python - <<PY
import json
d=json.load(open("$OUT/q.bench.json"))
print({k:d.get(k) for k in ("value","ms_per_step")})
for r in ("roofline","roofline_other"): print("   ", d[r]["kernel"][:40], d[r]["avg_launch_ms"], round(d[r]["frac"],4))
PY
