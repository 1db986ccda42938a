#!/usr/bin/env python3
"""Dispatch-by-dispatch timeline of the LAST step in a rocprofv3 rocpd database: start offset, duration and the idle gap in front of every
kernel, plus the wall time between two marker kernels (default: the region from the first launch after `tail_fwd` to `tail_bwd`, i.e.
the render head forward + loss + render head backward).
usage: python scripts/rocpd_timeline.py <db> [first_marker [last_marker]]
"""
import re
import sqlite3
import sys


def main(path, first="tail_fwd", last="tail_bwd"):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = db.execute("select %s, start, end from kernels order by start" % name_col).fetchall()
    marks = [i for i, r in enumerate(rows) if "ray_knn" in r[0]]
    rows = rows[marks[-1]:]
    t0, prev_end = rows[0][1], rows[0][1]
    a = b = None
    busy = 0
    for i, (name, s, e) in enumerate(rows):
        name = re.sub(r"^void ", "", name.replace("(anonymous namespace)::", ""))
        name = re.sub(r"\(.*", "", name)[:90]
        print("%4d  +%9.1f us  %8.1f us  gap %6.1f  %s" % (i, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, name))
        if first in name and a is None:
            a = i
        if last in name:
            b = i
        prev_end = max(prev_end, e)
        busy += e - s
    print("# step: %d launches, %.1f us wall, %.1f us busy" % (len(rows), (prev_end - t0) / 1e3, busy / 1e3))
    if a is not None and b is not None and b > a + 1:
        seg = rows[a + 1:b]
        print("# between %s and %s: %d launches, %.1f us wall (%.1f us busy)" % (first, last, len(seg), (rows[b][1] - rows[a][2]) / 1e3,
                                                                               sum(e - s for _, s, e in seg) / 1e3))


if __name__ == "__main__":
    main(*sys.argv[1:4])
