#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --maxfail=6 -k "env_step or with_gc or golden_runs or vanilla_batch or (sampled_seeds and ValueSim-20)" > $OUT/p.pytest.log 2>&1
echo "pytest rc=$?"; tail -n 4 $OUT/p.pytest.log | cut -c1-300
timeout 600 python bench.py --no-cpu-baseline > $OUT/p.bench.json 2> $OUT/p.bench.err
python - <<PY
import json
d=json.load(open("$OUT/p.bench.json"))
print({k:d.get(k) for k in ("value","ms_per_step","last_sim_phase_kcycles")})
for r in ("roofline","roofline_other"): print("   ", d[r]["kernel"][:40], d[r]["avg_launch_ms"], round(d[r]["frac"],4))
PY
