#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python scripts/selfplay_online.py --minutes ${MINUTES:-9.5} --max-nodes 100000 --games 512 --sims 200 --train-every 50 --out $OUT/online_learning.jsonl > $OUT/online.log 2> $OUT/online.err
echo "rc=$?"; tail -n 3 $OUT/online.err | cut -c1-200; cat $OUT/online_learning.jsonl | cut -c1-400
