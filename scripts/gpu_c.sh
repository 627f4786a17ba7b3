#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --maxfail=12 -k "ValueSimC-12 or vanillac or state_injection or vanilla" > $OUT/c.pytest.log 2>&1
echo "pytest rc=$?"; tail -n 15 $OUT/c.pytest.log | cut -c1-300
timeout 900 python bench.py > $OUT/c.bench.json 2> $OUT/c.bench.err
echo "bench rc=$?"; tail -c 2500 $OUT/c.bench.json; tail -n 3 $OUT/c.bench.err
cd /tmp
rm -rf /tmp/prof_kt /tmp/prof_fetch /tmp/prof_write
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/c.prof_kt.json 2> $GRAFT_REPO_ROOT/$OUT/c.prof_kt.err
echo "kt rc=$?"
find /tmp/prof_kt -name "*stats*.csv" | head; 
for f in $(find /tmp/prof_kt -name "*kernel_stats.csv"); do cp $f $GRAFT_REPO_ROOT/$OUT/c.kernel_stats.csv; done
head -n 12 $GRAFT_REPO_ROOT/$OUT/c.kernel_stats.csv | cut -c1-250
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof_fetch -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/c.prof_fetch.err
echo "fetch rc=$?"
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/prof_write -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/c.prof_write.err
echo "write rc=$?"
cd $GRAFT_REPO_ROOT
KEY=$(python -c "import json;print(json.load(open('$OUT/c.prof_kt.json'))['config']['workload_key'])")
python scripts/pmc_traffic.py $OUT/c.pmc_traffic.json $OUT/c.pmc_traffic.csv /tmp/prof_fetch /tmp/prof_write --last 10000 --workload-key "$KEY" --command "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-trace --output-format csv -- python bench.py --no-cpu-baseline (one pass per counter; last 10000 launches = the timed window)"
cat $OUT/c.pmc_traffic.csv
