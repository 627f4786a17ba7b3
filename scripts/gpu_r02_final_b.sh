#!/bin/bash
# the final bench line of the round (roofline.traffic now from this build's PMC passes) and the full-pool regime
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 600 python bench.py > $OUT/i.bench.json 2> $OUT/i.bench.err ) 2>&1 | grep real
python -c "
import json; d=json.load(open('$OUT/i.bench.json')); print({k:d.get(k) for k in ('value','ms_per_step')}); print(d['roofline']['kernel'][:20], d['roofline']['avg_launch_ms'], d['roofline']['traffic'], d['roofline_other']['kernel'][:20], d['roofline_other']['avg_launch_ms'], d['roofline_other']['traffic']); print(d['cpu_baseline']['cores'], d['cpu_baseline']['value'])"
( time timeout 600 python bench.py --no-cpu-baseline --warmup 75 --steps 20 > $OUT/i.bench_76.json 2> $OUT/i.bench_76.err ) 2>&1 | grep real
python -c "
import json; d=json.load(open('$OUT/i.bench_76.json')); print({k:d.get(k) for k in ('value','ms_per_step','sims_per_sec','gc')})"
