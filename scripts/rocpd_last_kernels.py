#!/usr/bin/env python3
"""Launch-order timeline of the LAST `n` kernels of a traced run: start offset, duration, gap to the previous kernel's end.
Usage: rocpd_last_kernels.py results.db [n=60] [anchor kernel substring: start at its last occurrence instead]"""
import sqlite3
import sys

sys.path.insert(0, __file__.rsplit('/', 1)[0])
from rocpd_stats import short  # noqa: E402

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else "kernel_name"
rows = [(short(n).split("<")[0][:56], s, e) for n, s, e in db.execute(f"select {name_col}, start, end from kernels order by start")]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
if len(sys.argv) > 3:
    idx = [i for i, r in enumerate(rows) if sys.argv[3] in r[0]]
    rows = rows[idx[-1]:idx[-1] + n]
else:
    rows = rows[-n:]
t0, prev_end, busy = rows[0][1], rows[0][1], 0.0
for name, s, e in rows:
    print(f"{(s - t0) / 1e3:9.1f} us  {name:58s} {(e - s) / 1e3:8.1f} us   gap {(s - prev_end) / 1e3:7.1f}")
    busy += (e - s) / 1e3
    prev_end = max(prev_end, e)
print(f"kernels {len(rows)}, sum {busy:.1f} us, span {(prev_end - t0) / 1e3:.1f} us")
