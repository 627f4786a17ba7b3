#!/bin/bash
# kernel trace of the DistValueSim bench line (which library kernels the distributional head runs on)
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd /tmp; rm -rf /tmp/p_dist
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_dist -- python $R/bench.py --agent DistValueSim --sims 1000 --no-cpu-baseline --steady-steps 0 --warmup 2 --steps 5 > $R/$OUT/ev.dist_kt.json 2> $R/$OUT/ev.dist_kt.err; echo "dist kernel trace rc=$?"
cd $R; python scripts/kernel_stats.py /tmp/p_dist $OUT/ev.kernel_stats_dist.csv --last 4000; head -n 14 $OUT/ev.kernel_stats_dist.csv | cut -c1-200
