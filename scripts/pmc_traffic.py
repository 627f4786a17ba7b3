"""HBM traffic per launch from rocprofv3 counter passes (one `--pmc <COUNTER> --kernel-trace --output-format csv` run
per counter): python scripts/pmc_traffic.py <out.json> <out.csv> <dir FETCH_SIZE> <dir WRITE_SIZE>"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

out_json, out_csv, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
acc = defaultdict(lambda: defaultdict(list))
for d in dirs:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                name = re.sub(r"\(.*$", "", r["Kernel_Name"]).replace("void ", "").strip()
                acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
kernels, rows = {}, []
for name, cs in acc.items():
    if not name.startswith("tmcts"):
        continue
    k = {}
    for c, vals in cs.items():
        k[c + "_KB_mean"] = sum(vals) / len(vals)
        k[c + "_KB_last50_mean"] = sum(vals[-50:]) / len(vals[-50:])
        k["launches"] = len(vals)
        rows.append([name, c, len(vals), k[c + "_KB_mean"], k[c + "_KB_last50_mean"]])
    kernels[name] = k
doc = {
    "command": "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-trace --output-format csv -- python bench.py --steps 1 "
               "--warmup 0 --sims 200 --no-cpu-baseline (one pass per counter)",
    "workload": "4096 games x first 200 simulations of move 1 (short traces: mean length ~20), ValueSim, HIP value net",
    "units": "Counter values are KiB per dispatch (rocprofv3 FETCH_SIZE/WRITE_SIZE); bytes = value*1024. "
             "MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane) coalesced streaming "
             "reads; narrow accesses are uncalibrated. The tree kernel's accesses are 4-16 B scattered (no correction "
             "applied); the value-net kernels stream 16 B/lane (x2 applies to their reads).",
    "kernels": kernels,
}
with open(out_json, "w") as f:
    json.dump(doc, f, indent=1)
with open(out_csv, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "counter", "launches", "mean_KiB_per_launch", "last50_mean_KiB_per_launch"])
    w.writerows(rows)
print("kernels:", sorted(kernels))
