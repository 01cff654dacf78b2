#!/usr/bin/env python3
"""Kernels of the LAST training epoch of a traced full-graph run (from its last k_adam_multi back to the one before), in launch
order with durations; GEMM + split-K reduce pairs are what scripts/gemm_shapes_epoch.py lists.
Usage: rocpd_last_epoch.py results.db [back]    (back = 1: the last epoch; 3: the third from last -- the last one of a `run.py` call is
followed by the evaluation sweep, whose sort / scan kernels then sit in front of its Adam launch)"""
import sqlite3
import sys

sys.path.insert(0, __file__.rsplit('/', 1)[0])
from rocpd_stats import short  # noqa: E402

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else "kernel_name"
rows = [(short(n).split("<")[0][:60], s, e) for n, s, e in db.execute(f"select {name_col}, start, end from kernels order by start")]
ad = [i for i, r in enumerate(rows) if r[0] == "k_adam_multi"]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
seg = rows[ad[-back - 1] + 1:ad[-back] + 1]
tot = 0.0
for n, s, e in seg:
    tot += (e - s) / 1e3
    print(f"{n:62s} {(e - s) / 1e3:8.1f} us")
print(f"kernels {len(seg)}, sum {tot:.1f} us, span {(seg[-1][2] - seg[0][1]) / 1e3:.1f} us")
