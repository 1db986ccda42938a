#!/usr/bin/env python3
"""Per-kernel averages of the PMC counters in rocprofv3 rocpd databases (one counter pass per database).

usage: python scripts/rocpd_pmc.py results.db [more.db ...] [--keep name1,name2] [--top N] [--schema]

Prints the `--top` (default 40) largest (kernel, counter) rows by summed value, and EVERY row of a kernel whose name
contains one of the `--keep` substrings whatever its rank (round 1 lost the ray_knn rows by cutting at 40)."""
import re
import sqlite3
import sys


def rows_of(path):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    cnt_col = "counter_name" if "counter_name" in cols else "pmc_name"
    val_col = "value" if "value" in cols else "counter_value"
    return db.execute("select %s, %s, %s from counters_collection" % (name_col, cnt_col, val_col)).fetchall()


def main(argv):
    paths = [a for a in argv if a.endswith(".db")]
    keep = [k for a in argv if a.startswith("--keep=") for k in a[7:].split(",") if k]
    if "--keep" in argv:
        keep += argv[argv.index("--keep") + 1].split(",")
    top = int(argv[argv.index("--top") + 1]) if "--top" in argv else 40
    if "--schema" in argv:
        db = sqlite3.connect(paths[0])
        for v in ("pmc_events", "counters_collection", "pmc_info"):
            print(v, [r[1] for r in db.execute("pragma table_info(%s)" % v)])
            print("   ", db.execute("select * from %s limit 1" % v).fetchall())
        return
    agg = {}
    for path in paths:
        for name, cnt, val in rows_of(path):
            name = re.sub(r"^void ", "", name.replace("(anonymous namespace)::", ""))
            name = re.sub(r"\(.*", "", name)
            a = agg.setdefault((name, cnt), [0, 0.0])
            a[0] += 1
            a[1] += float(val)
    ranked = sorted(agg.items(), key=lambda kv: -kv[1][1])
    shown = set(k for k, _ in ranked[:top]) | set(k for k, _ in ranked if any(s in k[0] for s in keep))
    print("%-70s %-26s %8s %18s %18s" % ("kernel", "counter", "calls", "sum", "avg_per_launch"))
    for (name, cnt), (n, s) in ranked:
        if (name, cnt) in shown:
            print("%-70s %-26s %8d %18.1f %18.1f" % (name[:70], cnt, n, s, s / n))


if __name__ == "__main__":
    main(sys.argv[1:])
