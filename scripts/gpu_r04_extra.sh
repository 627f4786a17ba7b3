#!/bin/bash
# extra lines of the round: the steady-state window with the online leg under torch.distributed.run (one rank, RCCL process
# group: the exchange step really goes through the collective), ValueSimLP's steady-state window
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --no-cpu-baseline --online --warmup 75 --steps 20 --steady-steps 0 --others none > $OUT/x.online_steady.json 2> $OUT/x.online_steady.err; echo "online steady rc=$?"
python -c "
import json; d=json.load(open('$OUT/x.online_steady.json')); print(d['value'], d['ms_per_step'], d['exchange'], d['gc'])"
timeout 600 python bench.py --agent ValueSimLP --no-cpu-baseline --warmup 75 --steps 20 --steady-steps 0 --others none > $OUT/x.lp_steady.json 2> $OUT/x.lp_steady.err; echo "lp steady rc=$?"
python -c "
import json; d=json.load(open('$OUT/x.lp_steady.json')); print(d['value'], d['ms_per_step'], d['requests']['fraction_not_posted'], d['gc'])"
