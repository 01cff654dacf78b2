#!/usr/bin/env python3
"""PMC values per kernel DISPATCH (in launch order) from a rocprofv3 rocpd database: kernel,dispatch_order,counter,value.
scripts/plan_kernel_times.py launches the plan kernels once per chunk size, so the n-th dispatch of a kernel = the n-th size."""
import sqlite3
import sys

sys.path.insert(0, __file__.rsplit('/', 1)[0])
from rocpd_stats import short  # noqa: E402

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select kernel_name, counter_name, dispatch_id, value from counters_collection").fetchall()
agg = {}
for name, ctr, disp, val in rows:
    k = short(name)
    if not k.startswith("k_"):
        continue
    agg.setdefault((k, ctr), {}).setdefault(disp, 0.0)
    agg[(k, ctr)][disp] += float(val)
for (k, c), d in sorted(agg.items()):
    for order, disp in enumerate(sorted(d)):
        print(f"{k},{order},{c},{d[disp]:.0f}")
