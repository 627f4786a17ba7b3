#!/bin/bash
# Round-2 evidence on one box: default bench line (with the CPU baseline), rocprofv3 kernel trace + PMC passes of the same
# command, then the whole GPU suite (xdist workers share the GPU; the suite's time is mostly the CPU oracle's).
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( time timeout 600 python bench.py > $OUT/h.bench.json 2> $OUT/h.bench.err ) 2>&1 | grep real
python -c "
import json; d=json.load(open('$OUT/h.bench.json')); print({k:d.get(k) for k in ('value','ms_per_step')}); print(d['roofline']['kernel'][:20], d['roofline']['avg_launch_ms'], d['roofline_other']['kernel'][:20], d['roofline_other']['avg_launch_ms']); print(d['cpu_baseline']['cores'], d['cpu_baseline']['value'], d['cpu_baseline'].get('all_cores'))"
cd /tmp; rm -rf /tmp/p_kt /tmp/p_fetch /tmp/p_write /tmp/p_sq
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_kt -- python $R/bench.py --no-cpu-baseline > $R/$OUT/h.prof_kt.json 2> $R/$OUT/h.prof_kt.err; echo "kt rc=$?"
for f in $(find /tmp/p_kt -name "*kernel_stats.csv"); do cp $f $R/$OUT/h.kernel_stats.csv; done
python $R/scripts/kernel_stats.py /tmp/p_kt $R/$OUT/h.kernel_stats_timed_window.csv --last 10000
head -n 5 $R/$OUT/h.kernel_stats_timed_window.csv | cut -c1-200
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_fetch -- python $R/bench.py --no-cpu-baseline > /dev/null 2> $R/$OUT/h.prof_fetch.err; echo "fetch rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_write -- python $R/bench.py --no-cpu-baseline > /dev/null 2> $R/$OUT/h.prof_write.err; echo "write rc=$?"
cd $R
KEY=$(python -c "import json;print(json.load(open('$OUT/h.prof_kt.json'))['config']['workload_key'])")
python scripts/pmc_traffic.py $OUT/h.pmc_traffic.json $OUT/h.pmc_traffic.csv /tmp/p_fetch /tmp/p_write --last 10000 --workload-key "$KEY" --command "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-trace --output-format csv -- python bench.py --no-cpu-baseline (one pass per counter; averaged over the last 10000 launches of every kernel = the timed window, moves 6-25)"
grep "k_sim_step\|k_vn_\|k_fc_out" $OUT/h.pmc_traffic.csv
timeout 600 python -m pytest tests -m gpu -q -n 5 --maxfail=12 --durations=5 > $OUT/h.pytest.log 2>&1
echo "pytest rc=$?"; tail -n 12 $OUT/h.pytest.log | cut -c1-250
cd /tmp
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d /tmp/p_sq -- python $R/bench.py --no-cpu-baseline --steps 4 > /dev/null 2> $R/$OUT/h.prof_sq.err; echo "sq rc=$?"
cd $R
python scripts/pmc_traffic.py $OUT/h.pmc_sq.json $OUT/h.pmc_sq.csv /tmp/p_sq --last 2000 --workload-key "$KEY (4 timed moves)" --command "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_INSTS_VMEM_RD --kernel-trace -- python bench.py --no-cpu-baseline --steps 4 (values are raw counter sums per dispatch, not KiB)"
grep "k_sim_step" $OUT/h.pmc_sq.csv
timeout 200 python bench.py --agent ValueSimLP --steps 3 --warmup 2 --no-cpu-baseline > $OUT/h.bench_lp.json 2> $OUT/h.bench_lp.err; echo "lp rc=$?"; python -c "
import json; d=json.load(open('$OUT/h.bench_lp.json')); print({k:d.get(k) for k in ('value','ms_per_step','evaluated_states_per_sec','mean_trace_len')}); print(d['roofline']['kernel'][:30], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
