#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
for sp in 2 4; do
rm -rf /tmp/prof_tl
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --split $sp --steps 4 --warmup 3 > $GRAFT_REPO_ROOT/$OUT/j.split$sp.json 2> $GRAFT_REPO_ROOT/$OUT/j.split$sp.err
echo "split $sp rc=$?"
python $GRAFT_REPO_ROOT/scripts/timeline.py /tmp/prof_tl
python - <<PY
import json
d=json.load(open("$GRAFT_REPO_ROOT/$OUT/j.split$sp.json"))
print($sp, {k:d.get(k) for k in ("value","ms_per_step")})
for r in ("roofline","roofline_other"): print("   ", d[r]["kernel"][:40], d[r]["avg_launch_ms"])
PY
# a short excerpt of the timeline: 24 consecutive kernels in the middle
python - <<PY
import csv, glob
f = glob.glob("/tmp/prof_tl/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("tmcts") or "tmcts" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
mid = rows[len(rows)*3//4: len(rows)*3//4+24]
t0=int(mid[0]["Start_Timestamp"])
for r in mid:
    print("%8.1f %8.1f  q=%s  %s" % ((int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-t0)/1e3, r.get("Queue_Id","?"), r["Kernel_Name"].split("(")[0][-22:]))
PY
done
