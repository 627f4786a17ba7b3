#!/bin/bash
# the longest online self-play run of the round: ValueSimLP, 512 games x 200 sims, fits every 50 moves; the value net is saved
# (reference checkpoint format) every 10 rounds -> gpurun_out/t.checkpoint.pt
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
MIN=${MIN:-11}
timeout $((MIN*60+120)) python scripts/selfplay_online.py --minutes $MIN --max-nodes 100000 --games 512 --sims 200 --train-every 50 \
   --out $OUT/t.online_learning.jsonl --save $OUT/t.checkpoint.pt > $OUT/t.online.log 2>&1; echo "online rc=$?"
tail -n 2 $OUT/t.online.log | cut -c1-700; ls -la $OUT/t.checkpoint.pt
