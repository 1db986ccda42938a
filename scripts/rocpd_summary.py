#!/usr/bin/env python3
"""Per-kernel summary (calls, total, average, share) of a rocprofv3 rocpd SQLite database.

rocprofv3 7.2 writes `*_results.db` by default; this turns its `kernels` view into the same table
`--stats` prints, so the summary can be committed under profiles/ as plain text.
usage: python scripts/rocpd_summary.py gpurun_out/prof1/r01_results.db [min_start_fraction | last:N[:marker]]

`last:N` keeps the dispatches from the N-th last launch of the marker kernel (default ray_knn -- either form of the kernel --, the first
kernel of every step) onwards, i.e. the last N steps: warm-up steps (MIOpen's solver search runs seconds of
naive convolutions there) stay out of the table.
"""
import re
import sqlite3
import sys


def main(path, skip="0"):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = db.execute("select %s, start, end from kernels order by start" % name_col).fetchall()
    if not rows:
        print("no kernel dispatches recorded")
        return
    t0, t1 = rows[0][1], rows[-1][2]
    if skip.startswith("last:"):
        parts = skip.split(":")
        n, marker = int(parts[1]), (parts[2] if len(parts) > 2 else "ray_knn")
        marks = [s for name, s, e in rows if marker in name]
        cut = marks[-n] if len(marks) >= n else t0
        print("# last %d steps (from the %d-th last %s launch): %.3f ms of wall time" % (n, n, marker, (t1 - cut) / 1e6))
    else:
        cut = t0 + (t1 - t0) * float(skip)
    agg = {}
    for name, s, e in rows:
        if s < cut:
            continue
        name = re.sub(r"^void ", "", name.replace("(anonymous namespace)::", ""))
        name = re.sub(r"\(.*", "", name)
        a = agg.setdefault(name, [0, 0])
        a[0] += 1
        a[1] += e - s
    total = sum(v[1] for v in agg.values())
    print("%-100s %8s %12s %10s %7s" % ("kernel", "calls", "total_ms", "avg_us", "share"))
    for name, (n, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-100s %8d %12.3f %10.2f %6.2f%%" % (name[:100], n, ns / 1e6, ns / n / 1e3, 100.0 * ns / total))
    print("%-100s %8d %12.3f" % ("TOTAL (kernel time)", sum(v[0] for v in agg.values()), total / 1e6))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "0")
