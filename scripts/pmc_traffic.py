"""HBM traffic per launch from rocprofv3 counter passes (one `--pmc <COUNTER> --kernel-trace --output-format csv` run
per counter, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass):

    python scripts/pmc_traffic.py out.json out.csv DIR_FETCH_SIZE DIR_WRITE_SIZE [--last N] [--workload-key KEY] [--command CMD]

`--last N`: average over the last N launches of every kernel only (= the timed window of the profiled bench command,
N = steps x sims); `--workload-key`: bench.py's config.workload_key of the profiled command line - bench.py reports
`roofline.traffic` from this file only when its own key matches."""
import argparse
import csv
import glob
import json
import os
import re
from collections import defaultdict

ap = argparse.ArgumentParser()
ap.add_argument("out_json")
ap.add_argument("out_csv")
ap.add_argument("dirs", nargs="+")
ap.add_argument("--last", type=int, default=0)
ap.add_argument("--workload-key", default=None)
ap.add_argument("--command", default=None)
args = ap.parse_args()

acc = defaultdict(lambda: defaultdict(list))
for d in args.dirs:
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
        sel = vals[-args.last:] if args.last > 0 else vals
        k[c + "_KB_mean"] = sum(sel) / len(sel)
        k[c + "_KB_all_mean"] = sum(vals) / len(vals)
        k["launches"] = len(vals)
        k["launches_averaged"] = len(sel)
        rows.append([name, c, len(vals), len(sel), k[c + "_KB_mean"], k[c + "_KB_all_mean"]])
    kernels[name] = k
doc = {
    "command": args.command or "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-trace --output-format csv -- python bench.py ... "
                               "(one pass per counter)",
    "workload_key": args.workload_key,
    "units": "Counter values are KiB per dispatch (rocprofv3 FETCH_SIZE/WRITE_SIZE); bytes = value*1024. "
             "MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane) coalesced streaming "
             "reads; narrow accesses are uncalibrated. The tree kernel's accesses are 16 B scattered (no correction "
             "applied); the value-net kernels stream 16 B/lane (x2 applies to their reads).",
    "kernels": kernels,
}
with open(args.out_json, "w") as f:
    json.dump(doc, f, indent=1)
with open(args.out_csv, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "counter", "launches", "launches_averaged", "mean_KiB_per_launch", "all_launches_mean_KiB"])
    w.writerows(rows)
print("kernels:", sorted(kernels))
