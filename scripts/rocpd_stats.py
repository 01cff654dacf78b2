#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (.db) kernel trace: per-kernel calls / total / avg / min / max (us).
Usage: python scripts/rocpd_stats.py results.db [out.csv]"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    m = re.search(r"(k_[a-z0-9_]+|scan_[a-z_]+)", name)
    if m:
        t = re.search(r"ILi(\d+)E", name)
        return m.group(1) + (f"<{t.group(1)}>" if t else "")
    return name[:90]


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(f"select {name_col}, start, end from kernels").fetchall()
    agg = {}
    for name, s, e in rows:
        k = short(name)
        d = (e - s) / 1e3
        a = agg.setdefault(k, [0, 0.0, 1e30, 0.0])
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    lines = ["kernel,calls,total_us,avg_us,min_us,max_us,pct"]
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"{k},{a[0]},{a[1]:.1f},{a[1]/a[0]:.2f},{a[2]:.2f},{a[3]:.2f},{100*a[1]/tot:.2f}")
    text = "\n".join(lines)
    print(text)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")


if __name__ == "__main__":
    main()
