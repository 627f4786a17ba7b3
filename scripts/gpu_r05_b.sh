#!/bin/bash
# round 5, call B: kernel timelines of one / two sub-batches with the stock and the co-residency-shaped value net
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd $R
HEAD="--no-cpu-baseline --steady-steps 0 --others none"
tl() {  # name, env-lib, bench args
  local name=$1 lib=$2; shift 2
  cd /tmp; rm -rf /tmp/p_$name
  TETRIS_MCTS_LIB=$lib timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_$name -- python $R/bench.py $HEAD "$@" > $R/$OUT/b.$name.json 2> $R/$OUT/b.$name.err; echo "$name rc=$?"
  cd $R; python scripts/timeline.py /tmp/p_$name $OUT/b.$name.timeline.json > $OUT/b.$name.timeline.txt 2>&1; cat $OUT/b.$name.timeline.txt
  python -c "import json;d=json.load(open('$OUT/b.$name.json'));print('$name', d['value'], d['ms_per_step'])"
  # a slice of the raw trace (the last 4000 launches) for a closer look
  f=$(find /tmp/p_$name -name '*kernel_trace.csv' | head -n 1); (head -n 1 $f; tail -n 4000 $f) | cut -d, -f1-12 | gzip > $OUT/b.$name.trace_tail.csv.gz
}
STOCK=$R/tetris_mcts_amd/libtetris_mcts_hip.so
for v in "$@"; do case $v in
  split1) tl split1 $STOCK --warmup 5 --steps 10 ;;
  split2) tl split2 $STOCK --split 2 --warmup 5 --steps 10 ;;
  rowp2) tl rowp2 $R/build_variants/rowp.so --split 2 --warmup 5 --steps 10 ;;
  rowp1) tl rowp1 $R/build_variants/rowp.so --warmup 5 --steps 10 ;;
esac; done
