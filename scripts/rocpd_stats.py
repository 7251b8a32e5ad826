#!/usr/bin/env python3
"""Per-kernel summary (calls, total / average / min / max duration, share) from a rocprofv3 `*_results.db`
(`rocprofv3 --kernel-trace --stats` writes the rocpd SQLite format on this ROCm): prints CSV like the `kernel_stats.csv`
of earlier rocprofv3 versions.  usage: rocpd_stats.py results.db [out.csv]"""
import csv
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
                       f"from kernels group by {name_col} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    out = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
    out.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for n, c, tot, avg, mn, mx in rows:
        n = re.sub(r"\s+", " ", n)
        out.writerow([n, c, tot, round(avg, 1), round(100.0 * tot / total, 3), mn, mx])


if __name__ == "__main__":
    main()
