#!/bin/bash
# Round-6 GPU calls, one script: scripts/gpu_r06.sh <part> [...]   (each part is one `gpurun` call's worth; output under gpurun_out/r06.*,
# the files that are evidence are copied to profiles/r06_* afterwards)
#   gcq      the quick gate of a collector change: smoke, the collector tests, the benched-regime tests that live on collections
#   gc       every parity test that goes through collections
#   suite    the whole -m gpu suite as the driver runs it (one serial process) + smoke
#   new      this round's new tests (fused Yogi, graph-replayed fit, hand-off stress), then the headline window's kernels
#   steady   the two steady-state windows (moves 76-95; random-init and trained net): ms/move, waiting launches, catch-up launches
#   sweep    collector parameters on both steady-state windows: marking allowance, speculative threshold
#   tune     trained net: marking allowance x cost allowance, with the tree kernel's time per launch
#   bench    the driver's command line
#   prof     rocprofv3 kernel traces (headline window, both steady-state windows) and the PMC passes of the headline window
#   lp dist  BASELINE configs[2] / configs[4]: bench line + kernel trace + the PMC passes
#   vn       the gate of a value-net change: parity tests, kernel traces of the headline window and of ValueSimLP
#   stations the value net's kernels in time (fc1's workgroups, the convolution's waves): profiles/r06_valuenet_stations_*.json
#   timeline one k_sim_step launch dissected (steady state, both nets): start / end of every simulation wave and collector workgroup
#   online   the online self-play run (ValueSimLP, 512 games x 200 sims, fits every 50 moves), MIN minutes (default 11)
#   evalck   a checkpoint's play strength (same protocol, no training), MIN minutes (default 9), CKPT=<file>
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
CK=tetris_mcts_amd/checkpoints/value_net_online_r05.pt
HEAD="--no-cpu-baseline --steady-steps 0 --others none"
line() { python - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
g = d["gc"]
print(sys.argv[1].split("/")[-1], "%.2f M/s %.2f ms/move" % (d["value"] / 1e6, d["ms_per_step"]),
      "| %s %.4f ms frac %.3f | %s %.4f ms frac %.3f" % (d["roofline"]["kernel"][:12], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"],
                                                       d["roofline_other"]["kernel"][:12], d["roofline_other"]["avg_launch_ms"], d["roofline_other"]["frac"]),
      "| collections %d waiting %.2f catch-up/move %.1f marker launches %.1f blocks %.0f" % (
          g["collections"], g["launches_per_collection"] or 0, g["catchup_launches_per_move"], g["marker"]["launches_per_collection"] or 0,
          g["marker"]["blocks_per_collection"] or 0))
PY
}
prof_kt() {   # name, last, bench args...: per-kernel table of the last N launches
  local name=$1 last=$2; shift 2
  cd /tmp; rm -rf /tmp/p_$name
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$name -- python $R/bench.py "$@" > $R/$OUT/r06.$name.json 2> $R/$OUT/r06.$name.err; echo "$name kernel trace rc=$?"
  cd $R; python scripts/kernel_stats.py /tmp/p_$name $OUT/r06.kernel_stats_$name.csv --last $last; head -n 5 $OUT/r06.kernel_stats_$name.csv | cut -c1-60,150-400
}
prof_pmc() {  # name, last, bench args...
  local name=$1 last=$2; shift 2
  cd /tmp; rm -rf /tmp/p_${name}_f /tmp/p_${name}_w
  timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_${name}_f -- python $R/bench.py "$@" > /dev/null 2> $R/$OUT/r06.${name}_fetch.err; echo "$name fetch rc=$?"
  timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_${name}_w -- python $R/bench.py "$@" > /dev/null 2> $R/$OUT/r06.${name}_write.err; echo "$name write rc=$?"
  cd $R
  KEY=$(python -c "import json;print(json.load(open('$OUT/r06.$name.json'))['config']['workload_key'])")
  python scripts/pmc_traffic.py $OUT/r06.pmc_traffic_$name.json $OUT/r06.pmc_traffic_$name.csv /tmp/p_${name}_f /tmp/p_${name}_w --last $last --workload-key "$KEY" \
    --command "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-trace --output-format csv -- python bench.py $* (one pass per counter; averaged over the last $last launches of every kernel = the timed window)"
  grep "k_sim_step\|k_vn_\|k_dn_" $OUT/r06.pmc_traffic_$name.csv
}
cd $R
for p in "$@"; do case $p in
gcq)
  timeout 300 python __graft_entry__.py smoke > $OUT/r06.smoke.log 2>&1; echo "smoke rc=$?"; tail -n 1 $OUT/r06.smoke.log | cut -c1-300
  ( time timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_collector.py tests/test_gpu_benched_regime.py -k "collector or waiting or grid or catch_up or steady_state or trained or full_pool or under_load" > $OUT/r06.gcq_tests.log 2>&1 ) 2>&1 | grep real
  tail -n 6 $OUT/r06.gcq_tests.log | cut -c1-220 ;;
gc)
  timeout 300 python __graft_entry__.py smoke > $OUT/r06.smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $OUT/r06.smoke.log | cut -c1-300
  ( time timeout 1500 python -m pytest -x -q -m gpu tests/test_gpu_tree.py tests/test_gpu_collector.py tests/test_gpu_benched_regime.py tests/test_gpu_dist_agent.py --durations=10 > $OUT/r06.gc_tests.log 2>&1 ) 2>&1 | grep real
  tail -n 30 $OUT/r06.gc_tests.log | cut -c1-220 ;;
suite)
  # (TM_TEST_FULL=1: the online training-set parity replays all 900 reference moves under all four policies - the driver's run: two of them)
  ( time TM_TEST_FULL=1 timeout 1700 python -m pytest tests -m gpu -q --durations=15 > $OUT/r06.pytest.log 2>&1 ) 2>&1 | grep real
  tail -n 24 $OUT/r06.pytest.log | cut -c1-200
  timeout 300 python __graft_entry__.py smoke > $OUT/r06.smoke.log 2>&1; echo "smoke rc=$?"; tail -n 1 $OUT/r06.smoke.log | cut -c1-300 ;;
new)
  ( time timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_train_dist.py tests/test_gpu_valuenet.py tests/test_gpu_dist_agent.py tests/test_gpu_tree.py -k "yogi or fit or training or stress or valuenet or head or online_training_loop" --durations=8 > $OUT/r06.new_tests.log 2>&1 ) 2>&1 | grep real
  tail -n 16 $OUT/r06.new_tests.log | cut -c1-220
  timeout 300 python bench.py $HEAD > $OUT/r06.head.json 2> $OUT/r06.head.err; echo "head rc=$?"; line $OUT/r06.head.json ;;
steady)
  timeout 600 python bench.py --no-cpu-baseline --others none --warmup 75 --steps 20 --steady-steps 0 > $OUT/r06.steady_random.json 2> $OUT/r06.steady_random.err; echo "random rc=$?"
  line $OUT/r06.steady_random.json
  timeout 600 python bench.py --checkpoint $CK --no-cpu-baseline --others none --warmup 75 --steps 20 --steady-steps 0 > $OUT/r06.steady_trained.json 2> $OUT/r06.steady_trained.err; echo "trained rc=$?"
  line $OUT/r06.steady_trained.json ;;
sweep)
  for sl in 60000 100000 150000 220000; do for net in random trained; do
    ck=""; [ $net = trained ] && ck="--checkpoint $CK"
    timeout 300 python bench.py $ck --no-cpu-baseline --others none --warmup 75 --steps 20 --steady-steps 0 --gc-slice-cycles $sl > $OUT/r06.sweep_slice${sl}_$net.json 2> /dev/null
    line $OUT/r06.sweep_slice${sl}_$net.json
  done; done
  for sp in 128 512 1024; do for net in random trained; do
    ck=""; [ $net = trained ] && ck="--checkpoint $CK"
    timeout 300 python bench.py $ck --no-cpu-baseline --others none --warmup 75 --steps 20 --steady-steps 0 --gc-spec-nodes $sp > $OUT/r06.sweep_spec${sp}_$net.json 2> /dev/null
    line $OUT/r06.sweep_spec${sp}_$net.json
  done; done ;;
tune)
  for cfg in "120000 0" "135000 0" "150000 8" "150000 16" "135000 16" "170000 0"; do set -- $cfg
    timeout 300 python bench.py --checkpoint $CK --no-cpu-baseline --others none --warmup 75 --steps 20 --steady-steps 0 --gc-slice-cycles $1 --gc-cost-units $2 > $OUT/r06.tune_$1_$2.json 2> /dev/null
    line $OUT/r06.tune_$1_$2.json
  done ;;
bench)
  ( time timeout 900 python bench.py > $OUT/r06.bench.json 2> $OUT/r06.bench.err ) 2>&1 | grep real
  line $OUT/r06.bench.json
  python -c "import json;d=json.load(open('$OUT/r06.bench.json'));print({k:v for k,v in d.items() if not isinstance(v,(dict,list))})" ;;
prof)
  prof_kt head 10000 $HEAD
  prof_pmc head 10000 $HEAD
  prof_kt steady_random 10000 --no-cpu-baseline --steady-steps 0 --others none --warmup 75 --steps 20
  prof_kt steady_trained 10000 --checkpoint $CK --no-cpu-baseline --steady-steps 0 --others none --warmup 75 --steps 20 ;;
lp)
  prof_kt lp 10000 --agent ValueSimLP $HEAD; line $OUT/r06.lp.json
  prof_pmc lp 10000 --agent ValueSimLP $HEAD ;;
dist)
  prof_kt dist 5000 --agent DistValueSim --sims 1000 --warmup 2 --steps 5 $HEAD; line $OUT/r06.dist.json
  prof_pmc dist 5000 --agent DistValueSim --sims 1000 --warmup 2 --steps 5 $HEAD ;;
ldsprobe)
  # the LDS bank conflicts of k_vn_conv's conv2 reads, MEASURED by removing them: a variant whose reads go to pitch-6 addresses
  # (32 consecutive floats a half-wave: conflict-free; the results are wrong, the instruction stream is the same) against the product,
  # same inputs: kernel time and SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  bash scripts/build_variant.sh conv_nc valuenet_conv.inc 's/boff2\[t\]\[j\] = y \* 8 + x + koff2\[j\];/boff2[t][j] = y * 6 + x + koff2[j];/' | tail -n 1
  for lib in product conv_nc; do for B in 1867 7169; do
    L=$R/tetris_mcts_amd/libtetris_mcts_hip.so; [ $lib != product ] && L=$R/build_variants/$lib.so
    cd /tmp; rm -rf /tmp/pp_k /tmp/pp_c
    TETRIS_MCTS_LIB=$L timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp_k -- python $R/scripts/conv_probe.py $B 200 > $R/$OUT/r06.ldsprobe_${lib}_$B.log 2>&1
    TETRIS_MCTS_LIB=$L timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d /tmp/pp_c -- python $R/scripts/conv_probe.py $B 50 > /dev/null 2>&1
    cd $R; python scripts/kernel_stats.py /tmp/pp_k $OUT/r06.ldsprobe_ks_${lib}_$B.csv --last 150 > /dev/null
    python scripts/pmc_traffic.py $OUT/r06.ldsprobe_pmc_${lib}_$B.json $OUT/r06.ldsprobe_pmc_${lib}_$B.csv /tmp/pp_c --last 40 --workload-key "conv_probe $B" --command "rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace -- python scripts/conv_probe.py $B 50 (raw counter sums per dispatch)" > /dev/null
    echo "== $lib B=$B"; grep "k_vn_" $OUT/r06.ldsprobe_ks_${lib}_$B.csv | cut -c1-40,150-260; grep "k_vn_conv" $OUT/r06.ldsprobe_pmc_${lib}_$B.csv | cut -c1-300
  done; done ;;
online)
  MIN=${MIN:-11}
  timeout $((MIN*60+120)) python scripts/selfplay_online.py --minutes $MIN --max-nodes 100000 --games 512 --sims 200 --train-every 50 \
     --out $OUT/r06.online_learning.jsonl --save $OUT/r06.checkpoint.pt > $OUT/r06.online.log 2>&1; echo "online rc=$?"
  tail -n 2 $OUT/r06.online.log | cut -c1-900; ls -la $OUT/r06.checkpoint.pt ;;
evalck)
  MIN=${MIN:-9}; CKPT=${CKPT:-$CK}
  timeout $((MIN*60+120)) python scripts/selfplay_online.py --minutes $MIN --max-nodes 100000 --games 512 --sims 200 --train-every 250 \
     --load $CKPT --no-train --out $OUT/r06.checkpoint_play.jsonl > $OUT/r06.checkpoint_play.log 2>&1; echo "rc=$?"
  tail -n 1 $OUT/r06.checkpoint_play.log | cut -c1-600 ;;
vn)
  # the gate of a value-net change: its parity tests, then the headline window's and ValueSimLP's kernel traces
  timeout 600 python -m pytest -x -q -m gpu tests/test_gpu_valuenet.py > $OUT/r06.vn_tests.log 2>&1; tail -n 2 $OUT/r06.vn_tests.log
  prof_kt head 10000 $HEAD; line $OUT/r06.head.json
  prof_kt lp 10000 --agent ValueSimLP $HEAD; line $OUT/r06.lp.json ;;
stations)
  # the value net's kernels in time: every k_vn_fc1 workgroup's and every k_vn_conv wave's stations (a -DTM_FC1_TIMELINE build)
  bash scripts/build_variant.sh fc1tl valuenet.hip '1i #define TM_FC1_TIMELINE 1' | tail -n 1
  for B in 1867 8960; do
    TETRIS_MCTS_LIB=$R/build_variants/fc1tl.so timeout 300 python scripts/fc1_timeline.py $B $OUT/r06.fc1_timeline_$B.json 2>&1 | tail -n 1 | cut -c1-600
  done ;;
timeline)
  # one launch dissected: when its simulation waves and its collector workgroups start and end (a -DTM_TIMELINE build)
  bash scripts/build_variant.sh timeline tree.hip '1i #define TM_TIMELINE 1' | tail -n 1
  TETRIS_MCTS_LIB=$R/build_variants/timeline.so timeout 600 python scripts/launch_timeline.py --out $OUT/r06.launch_timeline_random.json > $OUT/r06.launch_timeline_random.log 2>&1; echo "random rc=$?"
  TETRIS_MCTS_LIB=$R/build_variants/timeline.so timeout 600 python scripts/launch_timeline.py --checkpoint $CK --out $OUT/r06.launch_timeline_trained.json > $OUT/r06.launch_timeline_trained.log 2>&1; echo "trained rc=$?"
  TETRIS_MCTS_LIB=$R/build_variants/timeline.so timeout 600 python scripts/launch_timeline.py --checkpoint $CK --warm-moves 10 --moves 4 --out $OUT/r06.launch_timeline_trained_head.json > $OUT/r06.launch_timeline_trained_head.log 2>&1; echo "trained head rc=$?"
  # the same with the 32 032 bytes of LDS a workgroup had until the fix (26 granules of 1 280: four workgroups a CU)
  bash scripts/build_variant.sh timeline4 tree.hip '1i #define TM_TIMELINE 1' 's/    return (sim > gc ? sim : gc);/    return 32032;/' | tail -n 1
  TETRIS_MCTS_LIB=$R/build_variants/timeline4.so timeout 600 python scripts/launch_timeline.py --checkpoint $CK --moves 3 --out $OUT/r06.launch_timeline_trained_4_per_cu.json > $OUT/r06.launch_timeline_trained_4_per_cu.log 2>&1; echo "trained, 4 per CU rc=$?"
  tail -n 4 $OUT/r06.launch_timeline_random.log | cut -c1-1500; tail -n 4 $OUT/r06.launch_timeline_trained.log | cut -c1-1500; tail -n 2 $OUT/r06.launch_timeline_trained_head.log | cut -c1-1500 ;;
*) echo "unknown part $p" ;;
esac; done
