#!/bin/bash
# Round evidence: the default bench line (with the CPU baseline), rocprofv3 kernel trace + PMC passes of the same command.
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python bench.py > $OUT/f.bench.json 2> $OUT/f.bench.err
echo "bench rc=$?"; tail -c 1200 $OUT/f.bench.json; echo
cd /tmp; rm -rf /tmp/p_kt /tmp/p_fetch /tmp/p_write /tmp/p_sq
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_kt -- python $R/bench.py --no-cpu-baseline > $R/$OUT/f.prof_kt.json 2> $R/$OUT/f.prof_kt.err; echo "kt rc=$?"
for f in $(find /tmp/p_kt -name "*kernel_stats.csv"); do cp $f $R/$OUT/f.kernel_stats.csv; done
python $R/scripts/kernel_stats.py /tmp/p_kt $R/$OUT/f.kernel_stats_regs.csv
head -n 6 $R/$OUT/f.kernel_stats.csv | cut -c1-200
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_fetch -- python $R/bench.py --no-cpu-baseline > /dev/null 2> $R/$OUT/f.prof_fetch.err; echo "fetch rc=$?"
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_write -- python $R/bench.py --no-cpu-baseline > /dev/null 2> $R/$OUT/f.prof_write.err; echo "write rc=$?"
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d /tmp/p_sq -- python $R/bench.py --no-cpu-baseline --steps 4 > /dev/null 2> $R/$OUT/f.prof_sq.err; echo "sq rc=$?"
cd $R
KEY=$(python -c "import json;print(json.load(open('$OUT/f.prof_kt.json'))['config']['workload_key'])")
python scripts/pmc_traffic.py $OUT/f.pmc_traffic.json $OUT/f.pmc_traffic.csv /tmp/p_fetch /tmp/p_write --last 10000 --workload-key "$KEY" --command "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-trace --output-format csv -- python bench.py --no-cpu-baseline (one pass per counter; averaged over the last 10000 launches of every kernel = the timed window, moves 6-25)"
grep "k_sim_step\|k_vn_\|k_fc_out" $OUT/f.pmc_traffic.csv
python scripts/pmc_traffic.py $OUT/f.pmc_sq.json $OUT/f.pmc_sq.csv /tmp/p_sq --last 2000 --workload-key "$KEY (4 timed moves)" --command "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_INSTS_VMEM_RD --kernel-trace -- python bench.py --no-cpu-baseline --steps 4 (values are raw counter sums per dispatch, not KiB)"
grep "k_sim_step" $OUT/f.pmc_sq.csv
timeout 600 python bench.py --agent ValueSimLP --steps 3 --warmup 2 --no-cpu-baseline > $OUT/f.bench_lp.json 2> $OUT/f.bench_lp.err; echo "lp rc=$?"; python -c "
import json; d=json.load(open('$OUT/f.bench_lp.json')); print({k:d.get(k) for k in ('value','ms_per_step','evaluated_states_per_sec','mean_trace_len')}); print(d['roofline']['kernel'][:30], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
