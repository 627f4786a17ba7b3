#!/bin/bash
# what the driver runs at round end, in its own form: the GPU suite in one serial pytest process, smoke, the default bench line
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests/ -x -q -m gpu --durations=12 > $OUT/drv.pytest.log 2>&1 ) 2>&1 | grep real; echo "pytest rc=$?"; tail -n 20 $OUT/drv.pytest.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/drv.smoke.log 2>&1; echo "smoke rc=$?"; tail -n 1 $OUT/drv.smoke.log | cut -c1-300
( time timeout 900 python bench.py > $OUT/drv.bench.json 2> $OUT/drv.bench.err ) 2>&1 | grep real
python - <<PY
import json
d=json.load(open('$OUT/drv.bench.json'))
print({k:d.get(k) for k in ('value','ms_per_step')}, d['roofline']['frac'], d['roofline']['traffic'], d['roofline_other']['traffic'])
print('steady', d['steady_state']['ms_per_step'], d['steady_state']['value'])
for k,v in d['other_configs'].items(): print(k, v.get('value'), v.get('ms_per_step'), v.get('roofline',{}).get('traffic'), v.get('error'))
PY
