#!/bin/bash
# the remaining evidence of the round: configs[2] and configs[4] bench lines, the online self-play curve
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
MIN=${1:-4}
timeout 600 python bench.py --agent ValueSimLP --no-cpu-baseline --steady-steps 0 > $OUT/ev.bench_lp.json 2> $OUT/ev.bench_lp.err; echo "lp rc=$?"
timeout 600 python bench.py --agent DistValueSim --sims 1000 --no-cpu-baseline --steady-steps 0 > $OUT/ev.bench_dist.json 2> $OUT/ev.bench_dist.err; echo "dist rc=$?"
python - <<PY
import json
for n in ("lp","dist"):
    d=json.load(open("$OUT/ev.bench_%s.json"%n))
    print(n, {k:d.get(k) for k in ('value','ms_per_step','evaluated_states_per_sec')}, d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline_other']['avg_launch_ms'])
PY
timeout 900 python scripts/selfplay_online.py --minutes $MIN --max-nodes 100000 --games 512 --sims 200 --train-every 50 --out $OUT/ev.online_learning.jsonl > $OUT/ev.online.log 2>&1; echo "online rc=$?"
tail -n 3 $OUT/ev.online_learning.jsonl | cut -c1-600
