#!/usr/bin/env python3
"""Per-kernel PMC summary from a rocprofv3 rocpd database (counters_collection view).
Usage: python scripts/rocpd_pmc.py results.db [out.csv]   -> kernel,counter,dispatches,sum,avg_per_dispatch"""
import sqlite3
import sys
sys.path.insert(0, __file__.rsplit('/', 1)[0])
from rocpd_stats import short


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    rows = cur.execute("select kernel_name, counter_name, dispatch_id, value from counters_collection").fetchall()
    agg = {}
    for name, ctr, disp, val in rows:
        k = (short(name), ctr)
        a = agg.setdefault(k, [set(), 0.0])
        a[0].add(disp)
        a[1] += float(val)
    lines = ["kernel,counter,dispatches,sum,avg_per_dispatch"]
    for (k, c), (d, s) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if not k.startswith("k_") and not k.startswith("scan"):
            continue
        lines.append(f"{k},{c},{len(d)},{s:.0f},{s/len(d):.1f}")
    text = "\n".join(lines)
    print(text)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")


if __name__ == "__main__":
    main()
