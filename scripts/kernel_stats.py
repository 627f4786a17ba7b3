"""Per-kernel summary of a rocprofv3 `--kernel-trace --output-format csv` run:
python scripts/kernel_stats.py <dir with *_kernel_trace.csv> <out.csv>"""
import csv
import glob
import os
import sys
from collections import defaultdict

src, out = sys.argv[1], sys.argv[2]
rows = defaultdict(lambda: dict(n=0, tot=0, mn=1 << 62, mx=0, meta=None))
for path in glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True):
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            k = rows[r["Kernel_Name"]]
            k["n"] += 1
            k["tot"] += d
            k["mn"] = min(k["mn"], d)
            k["mx"] = max(k["mx"], d)
            k["meta"] = (r.get("VGPR_Count", ""), r.get("Accum_VGPR_Count", ""), r.get("SGPR_Count", ""),
                         r.get("LDS_Block_Size", ""), r.get("Scratch_Size", ""))
total = sum(k["tot"] for k in rows.values()) or 1
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage", "VGPR", "AGPR", "SGPR", "LDS",
                "Scratch"])
    for name, k in sorted(rows.items(), key=lambda kv: -kv[1]["tot"]):
        w.writerow([name, k["n"], k["tot"], round(k["tot"] / k["n"], 1), k["mn"], k["mx"], round(100.0 * k["tot"] / total, 3),
                    *k["meta"]])
print("kernels:", len(rows), "->", out)
