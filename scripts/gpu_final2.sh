#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python __graft_entry__.py smoke > $OUT/g.smoke.log 2>&1; echo "smoke rc=$?"
timeout 2400 python -m pytest tests -m gpu -q --maxfail=12 --durations=5 > $OUT/g.pytest.log 2>&1
echo "pytest rc=$?"; tail -n 12 $OUT/g.pytest.log | cut -c1-250
( time timeout 900 python bench.py > $OUT/g.bench.json 2> $OUT/g.bench.err ) 2>&1 | grep real
python -c "
import json; d=json.load(open('$OUT/g.bench.json')); print({k:d.get(k) for k in ('value','ms_per_step')}); print(d['roofline']['kernel'][:20], d['roofline']['traffic'], d['roofline_other']['kernel'][:20], d['roofline_other']['traffic']); print(d['cpu_baseline']['cores'], d['cpu_baseline']['value'], d['cpu_baseline']['all_cores'])"
cd /tmp; rm -rf /tmp/p_kt
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_kt -- python $R/bench.py --no-cpu-baseline > $R/$OUT/g.prof_kt.json 2> $R/$OUT/g.prof_kt.err; echo "kt rc=$?"
python $R/scripts/kernel_stats.py /tmp/p_kt $R/$OUT/g.kernel_stats_timed_window.csv --last 10000
head -n 5 $R/$OUT/g.kernel_stats_timed_window.csv | cut -c1-200
python -c "
import json; d=json.load(open('$R/$OUT/g.prof_kt.json')); print('profiled run events:', d['roofline']['kernel'][:20], d['roofline']['avg_launch_ms'], d['roofline_other']['kernel'][:20], d['roofline_other']['avg_launch_ms'])"
