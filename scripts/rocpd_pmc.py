#!/usr/bin/env python3
"""Per-kernel averages of the PMC counters in a rocprofv3 rocpd database (one counter pass per db).
usage: python scripts/rocpd_pmc.py results.db [--schema]"""
import re
import sqlite3
import sys


def main(path, schema=False):
    db = sqlite3.connect(path)
    if schema:
        for v in ("pmc_events", "counters_collection", "pmc_info"):
            print(v, [r[1] for r in db.execute("pragma table_info(%s)" % v)])
            print("   ", db.execute("select * from %s limit 1" % v).fetchall())
        return
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    cnt_col = "counter_name" if "counter_name" in cols else "pmc_name"
    val_col = "value" if "value" in cols else "counter_value"
    rows = db.execute("select %s, %s, %s from counters_collection" % (name_col, cnt_col, val_col)).fetchall()
    agg = {}
    for name, cnt, val in rows:
        name = re.sub(r"^void ", "", name.replace("(anonymous namespace)::", ""))
        name = re.sub(r"\(.*", "", name)
        a = agg.setdefault((name, cnt), [0, 0.0])
        a[0] += 1
        a[1] += float(val)
    print("%-70s %-14s %8s %16s %16s" % ("kernel", "counter", "calls", "sum", "avg_per_launch"))
    for (name, cnt), (n, s) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print("%-70s %-14s %8d %16.1f %16.1f" % (name[:70], cnt, n, s, s / n))


if __name__ == "__main__":
    main(sys.argv[1], "--schema" in sys.argv)
