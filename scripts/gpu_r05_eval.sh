#!/bin/bash
# the committed checkpoint's play strength: ValueSimLP, 512 games x 200 sims (the training run's protocol), no training, MIN minutes
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
MIN=${MIN:-9}
timeout $((MIN*60+120)) python scripts/selfplay_online.py --minutes $MIN --max-nodes 100000 --games 512 --sims 200 --train-every 250 \
   --load tetris_mcts_amd/checkpoints/value_net_online_r05.pt --no-train --out $OUT/t.checkpoint_play.jsonl > $OUT/t.checkpoint_play.log 2>&1; echo "rc=$?"
tail -n 1 $OUT/t.checkpoint_play.log | cut -c1-500
