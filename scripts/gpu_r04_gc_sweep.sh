#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for C in 80000 110000 150000 200000 260000; do
  timeout 300 python bench.py --no-cpu-baseline --others none --steady-steps 0 --warmup 75 --steps 20 --gc-slice-cycles $C > $OUT/s.gc_$C.json 2> $OUT/s.gc_$C.err
  python -c "
import json; d=json.load(open('$OUT/s.gc_$C.json')); print($C, round(d['ms_per_step'],2), d['gc']['catchup_launches_per_move'], round(d['gc']['launches_per_collection'],2), d['roofline']['avg_launch_ms'], d['roofline_other']['avg_launch_ms'], d['gc']['collections'])"
done
for SP in 128 512; do
  timeout 300 python bench.py --no-cpu-baseline --others none --steady-steps 0 --warmup 75 --steps 20 --gc-spec-nodes $SP > $OUT/s.sp_$SP.json 2> $OUT/s.sp_$SP.err
  python -c "
import json; d=json.load(open('$OUT/s.sp_$SP.json')); print('spec', $SP, round(d['ms_per_step'],2), d['gc']['catchup_launches_per_move'], round(d['gc']['launches_per_collection'],2), d['roofline']['avg_launch_ms'], d['roofline_other']['avg_launch_ms'])"
done
