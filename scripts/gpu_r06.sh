#!/bin/bash
# Round-6 GPU calls (one gpurun call each): scripts/gpu_r06.sh <part> [...]
#   gc       smoke + the parity tests that go through collections (the sweep marker's gate), first failure stops
#   gcq      the quick version of gc
#   steady   the steady-state windows (random-init and trained net), no CPU legs: ms/move, waiting launches, catch-up launches
#   suite    the whole -m gpu suite as the driver runs it + smoke
#   bench    the driver's command line
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
CK=tetris_mcts_amd/checkpoints/value_net_online_r05.pt
steady_line() { python - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for name, w in (("head", d), ("steady", d.get("steady_state"))):
    if not w: continue
    print(name, {k: w.get(k) for k in ("value", "ms_per_step", "gc", "tree_kernel_ms", "value_net_ms")})
PY
}
cd $R
for p in "$@"; do case $p in
gc)
  timeout 300 python __graft_entry__.py smoke > $OUT/r06.smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $OUT/r06.smoke.log | cut -c1-300
  ( time timeout 1500 python -m pytest -x -q -m gpu tests/test_gpu_tree.py tests/test_gpu_collector.py tests/test_gpu_benched_regime.py tests/test_gpu_dist_agent.py --durations=10 > $OUT/r06.gc_tests.log 2>&1 ) 2>&1 | grep real
  tail -n 30 $OUT/r06.gc_tests.log | cut -c1-220 ;;
gcq)
  # the quick gate: smoke, the collector tests, the benched-regime tests that live on collections
  timeout 300 python __graft_entry__.py smoke > $OUT/r06.smoke.log 2>&1; echo "smoke rc=$?"; tail -n 1 $OUT/r06.smoke.log | cut -c1-300
  ( time timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_collector.py tests/test_gpu_benched_regime.py -k "collector or waiting or grid or catch_up or steady_state or trained or full_pool or under_load" > $OUT/r06.gcq_tests.log 2>&1 ) 2>&1 | grep real
  tail -n 6 $OUT/r06.gcq_tests.log | cut -c1-220 ;;
steady)
  timeout 600 python bench.py --no-cpu-baseline --others none --warmup 75 --steps 20 --steady-steps 0 > $OUT/r06.steady_random.json 2> $OUT/r06.steady_random.err; echo "random rc=$?"
  steady_line $OUT/r06.steady_random.json
  timeout 600 python bench.py --checkpoint $CK --no-cpu-baseline --others none --warmup 75 --steps 20 --steady-steps 0 > $OUT/r06.steady_trained.json 2> $OUT/r06.steady_trained.err; echo "trained rc=$?"
  steady_line $OUT/r06.steady_trained.json ;;
sweep)
  # collector parameters on both steady-state windows: marking allowance, speculative threshold
  for sl in 60000 100000 150000 220000; do for net in random trained; do
    ck=""; [ $net = trained ] && ck="--checkpoint $CK"
    timeout 300 python bench.py $ck --no-cpu-baseline --others none --warmup 75 --steps 20 --steady-steps 0 --gc-slice-cycles $sl > $OUT/r06.sweep_slice${sl}_$net.json 2> /dev/null
    python -c "import json;d=json.load(open('$OUT/r06.sweep_slice${sl}_$net.json'));print('slice',$sl,'$net',round(d['ms_per_step'],2),round(d['gc']['launches_per_collection'],2),d['gc']['catchup_launches_per_move'])"
  done; done
  for sp in 128 512 1024; do for net in random trained; do
    ck=""; [ $net = trained ] && ck="--checkpoint $CK"
    timeout 300 python bench.py $ck --no-cpu-baseline --others none --warmup 75 --steps 20 --steady-steps 0 --gc-spec-nodes $sp > $OUT/r06.sweep_spec${sp}_$net.json 2> /dev/null
    python -c "import json;d=json.load(open('$OUT/r06.sweep_spec${sp}_$net.json'));print('spec',$sp,'$net',round(d['ms_per_step'],2),round(d['gc']['launches_per_collection'],2),d['gc']['catchup_launches_per_move'])"
  done; done ;;
tune)
  # trained-net steady state: marking allowance x cost allowance
  for cfg in "120000 0" "135000 0" "150000 8" "150000 16" "135000 16" "170000 0"; do set -- $cfg
    timeout 300 python bench.py --checkpoint $CK --no-cpu-baseline --others none --warmup 75 --steps 20 --steady-steps 0 --gc-slice-cycles $1 --gc-cost-units $2 > $OUT/r06.tune_$1_$2.json 2> /dev/null
    python -c "import json;d=json.load(open('$OUT/r06.tune_$1_$2.json'));print('slice',$1,'cost',$2,round(d['ms_per_step'],2),'tree',round(d['roofline']['avg_launch_ms'],4),'waiting',round(d['gc']['launches_per_collection'],2),'catch-up',d['gc']['catchup_launches_per_move'])"
  done ;;
stop)
  # when the marking of a launch ends: sixteenths of the simulation workgroups finished (tree.hip GC_STOP_DONE_16THS), built on the box
  for th in 4 8 12; do
    bash scripts/build_variant.sh stop$th "s/^constexpr int GC_STOP_DONE_16THS = [0-9]*;/constexpr int GC_STOP_DONE_16THS = $th;/" > /dev/null
    for net in random trained; do
      ck=""; [ $net = trained ] && ck="--checkpoint $CK"
      TETRIS_MCTS_LIB=$R/build_variants/stop$th.so timeout 300 python bench.py $ck --no-cpu-baseline --others none --warmup 75 --steps 20 --steady-steps 0 > $OUT/r06.stop${th}_$net.json 2> /dev/null
      python -c "import json;d=json.load(open('$OUT/r06.stop${th}_$net.json'));print('stop at',$th,'/16','$net',round(d['ms_per_step'],2),round(d['gc']['launches_per_collection'],2),d['gc']['catchup_launches_per_move'])"
    done; done ;;
suite)
  ( time timeout 1700 python -m pytest tests -m gpu -q --durations=15 > $OUT/r06.pytest.log 2>&1 ) 2>&1 | grep real
  tail -n 24 $OUT/r06.pytest.log | cut -c1-200
  timeout 300 python __graft_entry__.py smoke > $OUT/r06.smoke.log 2>&1; echo "smoke rc=$?"; tail -n 1 $OUT/r06.smoke.log | cut -c1-300 ;;
bench)
  ( time timeout 900 python bench.py > $OUT/r06.bench.json 2> $OUT/r06.bench.err ) 2>&1 | grep real
  steady_line $OUT/r06.bench.json ;;
esac; done
