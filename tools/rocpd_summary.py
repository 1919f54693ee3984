#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (--kernel-trace --stats) into a per-kernel table:
calls, total ms, avg us, % of GPU kernel time.  Usage: python tools/rocpd_summary.py <results.db> [skip_first_fraction]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)
    return name if len(name) <= 150 else name[:147] + "..."


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    if not rows:
        print("no kernels"); return
    t0, t1 = rows[0][1], rows[-1][2]
    agg = {}
    for n, s, e in rows:
        a = agg.setdefault(n, [0, 0, 1 << 62, 0])
        a[0] += 1; a[1] += e - s; a[2] = min(a[2], e - s); a[3] = max(a[3], e - s)
    tot = sum(v[1] for v in agg.values())
    print(f"# {db}: {len(rows)} kernel dispatches, {len(agg)} distinct kernels, GPU kernel time {tot/1e6:.2f} ms, wall span {(t1-t0)/1e6:.2f} ms")
    print(f"{'calls':>7} {'total_ms':>10} {'avg_us':>9} {'min_us':>8} {'max_us':>8} {'pct':>6}  kernel")
    for n, (cnt, ns, mn, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
        print(f"{cnt:7d} {ns/1e6:10.3f} {ns/cnt/1e3:9.2f} {mn/1e3:8.1f} {mx/1e3:8.1f} {100*ns/tot:6.2f}  {short(n)}")


if __name__ == "__main__":
    main()
