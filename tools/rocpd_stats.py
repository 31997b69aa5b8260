"""rocprofv3 (ROCm 7.2 default output = rocpd sqlite) -> the `--stats` kernel summary as CSV:
Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs.   python tools/rocpd_stats.py <results.db> [out.csv]"""
import csv
import sqlite3
import sys


def kernel_stats(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = cur.execute("select %s, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
                       "from kernels group by %s order by 3 desc" % (name, name)).fetchall()
    tot = float(sum(r[2] for r in rows)) or 1.0
    return [(r[0], r[1], r[2], r[3], 100.0 * r[2] / tot, r[4], r[5]) for r in rows]


if __name__ == "__main__":
    st = kernel_stats(sys.argv[1])
    out = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
    out.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in st:
        out.writerow([r[0], r[1], r[2], "%.1f" % r[3], "%.3f" % r[4], r[5], r[6]])
