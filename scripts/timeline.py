"""Summarise a rocprofv3 kernel_trace.csv of a bench run: per-kernel durations over the last three quarters of the launches, and how
much of the window is covered by 0 / 1 / 2+ kernels at once (do the sub-batches' kernels run concurrently or take turns?).
    python scripts/timeline.py <dir with *_kernel_trace.csv> [out.json]"""
import collections
import csv
import glob
import json
import re
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "tmcts" in r["Kernel_Name"]]


def short(name):
    name = re.sub(r"^void\s+", "", name).split("(")[0]
    return re.sub(r"<.*", "", name.split("::")[-1])


ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows)
ev = ev[len(ev) // 4:]
agg = collections.defaultdict(list)
for s, e, n, q in ev:
    agg[n].append(e - s)
out = {"kernels": {}}
for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    if len(v) < 50:
        continue
    out["kernels"][n] = dict(n=len(v), avg_us=sum(v) / len(v) / 1e3, max_us=max(v) / 1e3)
    print("%-16s n=%6d avg=%8.1f us  max=%8.1f" % (n, len(v), sum(v) / len(v) / 1e3, max(v) / 1e3))
# coverage: time with k kernels active
pts = []
for s, e, n, q in ev:
    pts.append((s, 1))
    pts.append((e, -1))
pts.sort()
cover = collections.defaultdict(int)
act, last = 0, pts[0][0]
for t, d in pts:
    cover[min(act, 2)] += t - last
    last, act = t, act + d
span = ev[-1][1] - ev[0][0]
tree = [(s, e) for s, e, n, q in ev if n == "k_sim_step"]
nn = sorted((s, e) for s, e, n, q in ev if n.startswith("k_vn") or n.startswith("k_dn"))
ov = tot = 0
j0 = 0
for s, e in tree:
    tot += e - s
    while j0 < len(nn) and nn[j0][1] <= s:
        j0 += 1
    j = j0
    while j < len(nn) and nn[j][0] < e:
        ov += max(0, min(e, nn[j][1]) - max(s, nn[j][0]))
        j += 1
out.update(span_ms=span / 1e6, idle=cover[0] / span, one_kernel=cover[1] / span, two_or_more=cover[2] / span,
           tree_time_overlapped_with_evaluator=ov / max(tot, 1), sum_kernel_time_over_span=sum(e - s for s, e, n, q in ev) / span,
           queues=sorted(set(q for _, _, _, q in ev)))
print("span %.1f ms: idle %.1f%%, one kernel %.1f%%, two or more %.1f%%; tree time overlapped with evaluator kernels %.1f%%; "
      "sum(kernel time)/span %.2f; queues %s" % (span / 1e6, 100 * out["idle"], 100 * out["one_kernel"], 100 * out["two_or_more"],
                                                  100 * out["tree_time_overlapped_with_evaluator"], out["sum_kernel_time_over_span"], out["queues"]))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
