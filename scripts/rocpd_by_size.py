#!/usr/bin/env python3
"""Per-kernel durations of a scripts/plan_kernel_times.py trace, one column per build (a build = one k_expand launch).
Usage: python scripts/rocpd_by_size.py results.db"""
import sqlite3
import sys

sys.path.insert(0, __file__.rsplit('/', 1)[0])
from rocpd_stats import short  # noqa: E402

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
ev = [(short(n).split("<")[0], s, e) for n, s, e in rows]
starts = [i for i, x in enumerate(ev) if x[0] == "k_expand"]
names = ["k_expand", "k_gather1c", "k_combine1_reset", "k_seg_transpose", "k_tile_counts", "k_build_groups", "k_gather2_items",
         "k_gather2_combine"]
print("build," + ",".join(names) + ",span_us")
for bi, a in enumerate(starts):
    b = starts[bi + 1] if bi + 1 < len(starts) else len(ev)
    win = [x for x in ev[a:b] if x[0] in names]
    d = {n: 0.0 for n in names}
    for x in win:
        d[x[0]] += (x[2] - x[1]) / 1e3
    span = (max(x[2] for x in win) - ev[a][1]) / 1e3
    print(f"{bi}," + ",".join(f"{d[n]:.1f}" for n in names) + f",{span:.1f}")
