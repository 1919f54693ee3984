#!/usr/bin/env python
"""Kernels of a rocprofv3 rocpd SQLite database (--kernel-trace) sorted by LAUNCH COUNT: launches per step, average duration, ms per step.
A launch costs ~5 us inside the replayed train-step graph whatever it does, so this is the to-do list for launch merging.

The trace of `bench.py` holds the eager warm-up steps (first-use work: weight packing, workspace fills, the capture pass) AND the replayed
steps; dividing everything by the step count charges the one-time work to every step.  So the steady state is found from the trace
itself: the smallest window length p whose last three windows of p dispatches hold the same multiset of kernel names is one replayed
step; counts and times are then taken over the trailing run of such windows only.  Falls back to total / <steps> when no period is found.
usage: python tools/rocpd_counts.py <results.db> <steps in the trace>"""
import collections
import sqlite3
import sys


def steady_period(names, lo=100, hi=4000, max_tail=600):
    """(period p, tail o): the dispatches [n - o - k p, n - o) are k replayed steps (equal multisets of kernel names per window of p); the
    last o dispatches are whatever ran after the last replay.  Multisets are compared through sums of per-name random 64-bit values
    (prefix sums: O(1) per window)."""
    import random
    rnd = random.Random(1234)
    val = {}
    pre = [0]
    for nm in names:
        if nm not in val:
            val[nm] = rnd.getrandbits(61)
        pre.append(pre[-1] + val[nm])
    n = len(names)
    for p in range(lo, min(hi, n // 4) + 1):
        for o in range(0, min(max_tail, n - 4 * p) + 1):
            e = n - o
            w = pre[e] - pre[e - p]
            if w == pre[e - p] - pre[e - 2 * p] == pre[e - 2 * p] - pre[e - 3 * p]:
                return p, o
    return 0, 0


def main():
    c = sqlite3.connect(sys.argv[1])
    steps = int(sys.argv[2])
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    names = [r[0] for r in rows]
    p, tail = steady_period(names)
    if p:
        e = len(names) - tail
        ref = collections.Counter(names[e - p:e])
        k = 1
        while (k + 1) * p <= e and collections.Counter(names[e - (k + 1) * p:e - k * p]) == ref:
            k += 1
        rows = rows[e - k * p:e]
        print(f"# steady state: {p} dispatches per replayed step (period found over the last {k} steps of the trace, {tail} dispatches after "
              f"them; {len(names)} dispatches in the whole trace = {len(names) / steps:.0f} per step with the warm-up's one-time work charged to "
              f"every step)")
        steps = k
    else:
        print(f"# no steady period found: {len(names)} dispatches / {steps} steps = {len(names) / steps:.0f} per step (warm-up included)")
    agg = {}
    for n, s, e in rows:
        a = agg.setdefault(n, [0, 0])
        a[0] += 1
        a[1] += e - s
    tot_ns = sum(v[1] for v in agg.values())
    print(f"# kernel time {tot_ns / steps / 1e6:.3f} ms per step; kernels sorted by launch count")
    for n, (k_, ns) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:60]:
        print(f"{k_ / steps:7.1f}/step {ns / k_ / 1e3:8.2f} us avg {ns / steps / 1e6:7.3f} ms/step  {n[:110]}")


if __name__ == "__main__":
    main()
