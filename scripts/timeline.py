"""Summarise a rocprofv3 kernel_trace.csv: per kernel name durations, and how much k_sim_step overlaps value-net kernels."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("tmcts")]
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows]
ev.sort()
agg = collections.defaultdict(list)
for s, e, n, q in ev:
    agg[n].append(e - s)
for n, v in agg.items():
    v = v[len(v) // 4:]
    print("%-14s n=%5d avg=%8.1f us" % (n, len(v), sum(v) / len(v) / 1e3))
tree = [(s, e) for s, e, n, q in ev if n == "k_sim_step"]
nn = [(s, e) for s, e, n, q in ev if n.startswith("k_vn") or n == "k_fc_out"]
# overlap of each tree kernel with any nn kernel
j = 0
ov = tot = 0
for s, e in tree[len(tree) // 4:]:
    tot += e - s
    for a, b in nn:
        if b <= s or a >= e:
            continue
        ov += min(e, b) - max(s, a)
span = ev[-1][1] - ev[len(ev) // 4][0]
busy = sum(e - s for s, e, n, q in ev[len(ev) // 4:])
print("tree time overlapped with NN kernels: %.1f%%   sum(kernel time)/span = %.2f  queues=%s" % (100.0 * ov / max(tot, 1), busy / span, sorted(set(q for _, _, _, q in ev))))
