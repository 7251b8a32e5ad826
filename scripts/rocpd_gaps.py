#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 kernel trace (`*_results.db`, rocpd SQLite): for every kernel
launch the gap to the END of its predecessor, grouped by the kernel that follows the gap.
usage: rocpd_gaps.py results.db [min_kernels_between_big_gaps]"""
import re
import sqlite3
import sys
from collections import defaultdict


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
    short = lambda n: re.sub(r"\(anonymous namespace\)::|void ", "", re.sub(r"\s+", " ", n))[:70]
    by = defaultdict(list)
    busy = 0
    for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
        g = s1 - e0
        busy += e1 - s1
        if g < 200_000:            # longer pauses are host-side (between clips, warm-up, synchronisations)
            by[short(n1)].append(g)
    tot = sum(sum(v) for v in by.values())
    print(f"kernels {len(rows)}, busy {busy / 1e6:.1f} ms, gaps < 200 us: {tot / 1e6:.2f} ms ({100 * tot / max(busy, 1):.1f} % of busy)")
    print(f"{'kernel after the gap':72s} {'n':>6s} {'median us':>10s} {'mean us':>9s} {'total ms':>9s}")
    for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:14]:
        v.sort()
        print(f"{k:72s} {len(v):6d} {v[len(v) // 2] / 1e3:10.2f} {sum(v) / len(v) / 1e3:9.2f} {sum(v) / 1e6:9.2f}")


if __name__ == "__main__":
    main()
