#!/bin/bash
# steady-state kernel trace: distribution of k_sim_step durations over the last moves
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/p_st
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_st -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --warmup 78 --steps 4 --steady-steps 0 > $GRAFT_REPO_ROOT/$OUT/rq.json 2> $GRAFT_REPO_ROOT/$OUT/rq.err; echo "rc=$?"
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, numpy as np
f = glob.glob("/tmp/p_st/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
sim = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)) for r in rows if "k_sim_step" in r["Kernel_Name"]]
sim.sort()
last = sim[-2400:]
d = np.array([x[1] for x in last]) / 1e3
g = np.array([x[2] for x in last])
print("k_sim_step last", len(d), "launches: mean %.1f us  p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % (d.mean(), *np.percentile(d, [10, 50, 90, 99]), d.max()))
print("grid sizes:", np.unique(g, return_counts=True))
full = d[g == g.max()]
print("full-grid launches", len(full), "mean %.1f p50 %.1f p90 %.1f" % (full.mean(), np.percentile(full, 50), np.percentile(full, 90)))
h, e = np.histogram(full, bins=[0, 40, 50, 60, 70, 80, 90, 100, 120, 150, 200, 400, 10000])
print("histogram us:", list(zip(e[:-1].tolist(), h.tolist())))
for name in ("k_vn_conv", "k_vn_fc1", "k_fc_out"):
    v = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows if name in r["Kernel_Name"])[-2400:]
    a = np.array([x[1] for x in v]) / 1e3
    print(name, "mean %.1f p50 %.1f p90 %.1f" % (a.mean(), np.percentile(a, 50), np.percentile(a, 90)))
PY
