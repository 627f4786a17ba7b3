"""Summarise a rocprofv3 rocpd database (`*_results.db`, the ROCm 7.2 default output of
`rocprofv3 --kernel-trace --stats`) into the per-kernel table `--stats` prints: calls, total, average,
min, max duration and share of GPU time.  Usage: python scripts/rocpd_stats.py results.db out.csv"""
import csv
import sqlite3
import sys


def main(db_path, out_csv):
    cur = sqlite3.connect(db_path).cursor()
    rows = cur.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), "
        "max(grid_x), max(workgroup_x) from kernels group by name order by 3 desc").fetchall()
    tot = float(sum(r[2] for r in rows))
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage", "VGPR", "AGPR",
                    "SGPR", "LDS", "Scratch", "GridX", "WorkgroupX"])
        for r in rows:
            w.writerow([r[0], r[1], r[2], "%.1f" % r[3], r[4], r[5], "%.3f" % (100.0 * r[2] / tot)] + list(r[6:]))
    for r in rows[:12]:
        print("%-70s calls=%6d avg_us=%9.2f pct=%6.2f" % (r[0][:70], r[1], r[3] / 1e3, 100.0 * r[2] / tot))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
