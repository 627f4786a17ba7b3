"""Per-kernel summary of a rocprofv3 `--kernel-trace --output-format csv` run:

    python scripts/kernel_stats.py <dir with *_kernel_trace.csv> <out.csv> [--last N]

`--last N`: statistics over the last N launches of every kernel only (in start order) - the timed window of the profiled
bench command (N = steps x sims), so the averages can be held against bench.py's roofline.avg_launch_ms."""
import csv
import glob
import os
import sys
from collections import defaultdict

args = [a for a in sys.argv[1:] if not a.startswith("--")]
last = int(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else 0
if "--last" in sys.argv:
    args.remove(sys.argv[sys.argv.index("--last") + 1])
src, out = args[0], args[1]
launches = defaultdict(list)
meta = {}
for path in glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True):
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            launches[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
            meta[r["Kernel_Name"]] = (r.get("VGPR_Count", ""), r.get("Accum_VGPR_Count", ""), r.get("SGPR_Count", ""),
                                      r.get("LDS_Block_Size", ""), r.get("Scratch_Size", ""))
rows = {}
for name, ls in launches.items():
    ls.sort()
    d = [x[1] for x in (ls[-last:] if last > 0 else ls)]
    rows[name] = dict(n=len(ls), used=len(d), tot=sum(d), mn=min(d), mx=max(d))
total = sum(k["tot"] for k in rows.values()) or 1
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "CallsAveraged", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage", "VGPR", "AGPR", "SGPR",
                "LDS", "Scratch"])
    for name, k in sorted(rows.items(), key=lambda kv: -kv[1]["tot"]):
        w.writerow([name, k["n"], k["used"], k["tot"], round(k["tot"] / k["used"], 1), k["mn"], k["mx"],
                    round(100.0 * k["tot"] / total, 3), *meta[name]])
print("kernels:", len(rows), "->", out)
