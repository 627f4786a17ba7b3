#!/bin/bash
# SQ counters of the convolution kernel (where its wave cycles go)
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd /tmp; rm -rf /tmp/p_cv
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d /tmp/p_cv -- python $R/bench.py --no-cpu-baseline --steady-steps 0 --warmup 1 --steps 2 > /dev/null 2> $R/$OUT/ev.cv_sq.err; echo "rc=$?"
cd $R; python scripts/pmc_traffic.py $OUT/ev.pmc_sq_conv.json $OUT/ev.pmc_sq_conv.csv /tmp/p_cv --last 800 --workload-key "conv sq" --command "rocprofv3 --pmc SQ_* -- python bench.py --warmup 1 --steps 2 (raw counter sums per dispatch)"
grep "k_vn_conv\|k_vn_fc1" $OUT/ev.pmc_sq_conv.csv
python scripts/kernel_stats.py /tmp/p_cv $OUT/ev.pmc_sq_conv_durations.csv --last 800; head -n 4 $OUT/ev.pmc_sq_conv_durations.csv | cut -c1-120
