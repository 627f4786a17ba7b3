#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 500 python scripts/selfplay_online.py --minutes 6 --max-nodes 100000 --games 512 --sims 200 --train-every 50 --out $OUT/ev.online_learning_6min.jsonl > $OUT/ev.online6.log 2>&1; echo "online rc=$?"; tail -n 2 $OUT/ev.online6.log | cut -c1-600
