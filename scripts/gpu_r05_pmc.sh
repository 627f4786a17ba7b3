#!/bin/bash
# memory-system counters of the tree kernel (headline workload, 2 timed moves per pass): where does a dependent round trip's time go?
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
HEAD="--no-cpu-baseline --steady-steps 0 --others none --warmup ${W:-4} --steps 2"
pass() {  # name, counters...
  local name=$1; shift
  cd /tmp; rm -rf /tmp/q_$name
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/q_$name -- python $R/bench.py $HEAD > /dev/null 2> $R/$OUT/pmc.$name.err; echo "$name rc=$?"
  cd $R; python scripts/pmc_traffic.py $OUT/pmc.$name.json $OUT/pmc.$name.csv /tmp/q_$name --last 1000 --workload-key "head (2 timed moves)" --command "rocprofv3 --pmc $* --kernel-trace -- python bench.py $HEAD (raw counter sums per dispatch)" > /dev/null
  grep "k_sim_step" $OUT/pmc.$name.csv | cut -c1-400
}
pass tlb TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum
pass lat TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum
pass l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum
pass stall TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE
head -n 1 $OUT/pmc.tlb.csv | cut -c1-300
