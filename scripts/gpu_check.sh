#!/bin/bash
# One GPU-box session: smoke, the gpu test suite, bench variants.  Everything lands in gpurun_out/$TAG.*
TAG=${1:-a}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
echo "== smoke" | tee $OUT/$TAG.smoke.log
timeout 600 python __graft_entry__.py smoke >> $OUT/$TAG.smoke.log 2>&1
echo "smoke rc=$?" | tee -a $OUT/$TAG.smoke.log
tail -n 5 $OUT/$TAG.smoke.log
echo "== pytest gpu"
timeout 2400 python -m pytest tests -m gpu -q --maxfail=10 -x --durations=15 > $OUT/$TAG.pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/$TAG.pytest.log
tail -n 40 $OUT/$TAG.pytest.log
for sp in ${SPLITS:-1 2 4}; do
  echo "== bench split $sp"
  timeout 900 python bench.py --split $sp --no-cpu-baseline > $OUT/$TAG.bench_split$sp.json 2> $OUT/$TAG.bench_split$sp.err
  echo "rc=$?"; tail -c 1500 $OUT/$TAG.bench_split$sp.json; tail -n 3 $OUT/$TAG.bench_split$sp.err
done
