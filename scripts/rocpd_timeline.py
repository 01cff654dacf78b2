#!/usr/bin/env python3
"""Timeline of the LAST chunk of a traced `bench.py --steps 20` run: every plan kernel and the first / last dense kernels with start
and end relative to the chunk's first kernel, plus the idle gaps between consecutive kernels.  Usage: rocpd_timeline.py results.db"""
import sqlite3
import sys

sys.path.insert(0, __file__.rsplit('/', 1)[0])
from rocpd_stats import short  # noqa: E402

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else "kernel_name"
rows = [(short(n).split("<")[0], s, e) for n, s, e in db.execute(f"select {name_col}, start, end from kernels order by start")]
idx = [i for i, r in enumerate(rows) if r[0] == "k_expand"]
a = idx[-1]
seg = rows[a:]
# stop at the last dense kernel of the chunk (the XCD-resident chunk kernel, or the last k_grad_reduce of the launch chain)
last = max(i for i, r in enumerate(seg) if r[0] in ("k_grad_reduce", "k_train_chunk_xcd"))
seg = seg[:last + 1]
t0 = seg[0][1]
prev_end = t0
gaps = 0.0
for i, (n, s, e) in enumerate(seg):
    gap = (s - prev_end) / 1e3
    if gap > 0:
        gaps += gap
    if i < 14 or i >= len(seg) - 6:
        print(f"{n:22s} start {(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f} us  gap before {gap:6.1f}")
    elif i == 14:
        print("   ...")
    prev_end = max(prev_end, e)
print(f"chunk span {(seg[-1][2] - t0) / 1e3:.1f} us, kernels {len(seg)}, sum of gaps {gaps:.1f} us")
