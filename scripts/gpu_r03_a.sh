#!/bin/bash
# r03 call A: default line; sub-batch streams with the convolution capped at one workgroup per CU; the polled overlap build
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
summ() { python - "$1" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print(sys.argv[1].split('/')[-1], "ms/move", round(d["ms_per_step"],2), "tree", round(d["roofline_other"]["avg_launch_ms"],4), "nn", round(d["roofline"]["avg_launch_ms"],4), "err", d["error_games"], "len", round(d["mean_trace_len"],3), d["max_trace_len"], "exp/s", round(d["value"]), "catchup", d["gc"]["catchup_launches"])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
timeout 200 python bench.py --no-cpu-baseline > $OUT/ra.default.json 2> $OUT/ra.default.err; echo "default rc=$?"; summ $OUT/ra.default.json
for cap in 1 2; do for sp in 2 4; do
  TM_CONV_WG_CAP=$cap timeout 200 python bench.py --no-cpu-baseline --split $sp > $OUT/ra.split${sp}_cap${cap}.json 2> $OUT/ra.split${sp}_cap${cap}.err; echo "split$sp cap$cap rc=$?"; summ $OUT/ra.split${sp}_cap${cap}.json
done; done
TM_CONV_WG_CAP=1 timeout 200 python bench.py --no-cpu-baseline --split 8 > $OUT/ra.split8_cap1.json 2> $OUT/ra.split8_cap1.err; echo "split8 cap1 rc=$?"; summ $OUT/ra.split8_cap1.json
TM_CONV_WG_CAP=1 timeout 200 python bench.py --no-cpu-baseline > $OUT/ra.split1_cap1.json 2> $OUT/ra.split1_cap1.err; echo "split1 cap1 rc=$?"; summ $OUT/ra.split1_cap1.json
# timeline of the best candidate (kernel trace, 3 timed moves)
cd /tmp; rm -rf /tmp/p_tl
TM_CONV_WG_CAP=1 timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_tl -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --split 4 --steps 3 --warmup 3 > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/ra.tl.err; echo "tl rc=$?"
cd $GRAFT_REPO_ROOT; python scripts/timeline.py /tmp/p_tl 2>&1 | tail -8
# the polled overlap build (never run before): smoke, then the bench under a hard time limit
export TETRIS_MCTS_LIB=$PWD/variants/libtetris_overlap.so
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/ra.overlap.smoke.log 2>&1; echo "overlap smoke rc=$?"; tail -n 3 $OUT/ra.overlap.smoke.log | cut -c1-300
timeout 150 python bench.py --no-cpu-baseline > $OUT/ra.overlap.json 2> $OUT/ra.overlap.err; echo "overlap bench rc=$?"; summ $OUT/ra.overlap.json; tail -n 3 $OUT/ra.overlap.err | cut -c1-300
