#!/bin/bash
# round 5, call A: the two-rank test, per-wave cycles for the row-P model, and the row-P experiment (two sub-batches on two
# streams with the co-residency-shaped value net, build_variants/rowp.so) with kernel timelines
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd $R
HEAD="--no-cpu-baseline --steady-steps 0 --others none"
( time timeout 900 python -m pytest tests/test_gpu_two_ranks.py -x -q > $OUT/a.two_ranks.log 2>&1 ) 2>&1 | grep real; tail -n 3 $OUT/a.two_ranks.log
timeout 300 python __graft_entry__.py smoke > $OUT/a.smoke.log 2>&1; tail -n 1 $OUT/a.smoke.log
timeout 300 python scripts/wave_cycles.py $OUT/a.wave_cycles.npz 2> $OUT/a.wave.err | tail -n 1
tl() {  # name, env-lib, bench args
  local name=$1 lib=$2; shift 2
  cd /tmp; rm -rf /tmp/p_$name
  TETRIS_MCTS_LIB=$lib timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_$name -- python $R/bench.py $HEAD "$@" > $R/$OUT/a.$name.json 2> $R/$OUT/a.$name.err; echo "$name rc=$?"
  cd $R; python scripts/timeline.py /tmp/p_$name > $OUT/a.$name.timeline.txt 2>&1; cat $OUT/a.$name.timeline.txt
  python -c "import json;d=json.load(open('$OUT/a.$name.json'));print('$name', d['value'], d['ms_per_step'])"
}
STOCK=$R/tetris_mcts_amd/libtetris_mcts_hip.so
tl split1 $STOCK --warmup 5 --steps 10
tl split2 $STOCK --split 2 --warmup 5 --steps 10
tl rowp2 $R/build_variants/rowp.so --split 2 --warmup 5 --steps 10
