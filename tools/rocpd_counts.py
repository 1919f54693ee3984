#!/usr/bin/env python
"""Kernels of a rocprofv3 rocpd SQLite database (--kernel-trace) sorted by LAUNCH COUNT: launches per step, average duration, ms per step.
A launch costs ~5 us inside the replayed train-step graph whatever it does, so this is the to-do list for launch merging.

The trace of `bench.py` holds the eager warm-up steps (first-use work: weight packing, workspace fills, the capture pass) AND the replayed
steps; dividing everything by the step count charges the one-time work to every step.  So the steady state is found from the trace
itself: the smallest window length p whose last three windows of p dispatches hold the same multiset of kernel names is one replayed
step; counts and times are then taken over the trailing run of such windows only.  Falls back to total / <steps> when no period is found.
usage: python tools/rocpd_counts.py <results.db> <steps in the trace> [--by-time]"""
import collections
import sqlite3
import sys


def steady_period(names):
    """(period p, tail o): the dispatches [n - o - k p, n - o) are k replayed steps.  The period is the spacing of the kernels that run
    ONCE per step (the optimizer's kernel, the loss kernels ...): for every kernel name with at least six launches whose last five
    spacings are equal that spacing is a vote, the most frequent vote wins, and the windows end right after the last launch of one of
    the voters.  (Comparing multisets of consecutive windows alone is not enough: a network of 24 identical blocks has windows shorter
    than a step, by whole blocks, that agree several times in a row.)"""
    pos = collections.defaultdict(list)
    for i, nm in enumerate(names):
        pos[nm].append(i)
    votes = collections.Counter()
    last = {}
    for nm, ps in pos.items():
        if len(ps) >= 6:
            d = [ps[-j] - ps[-j - 1] for j in range(1, 6)]
            if min(d) == max(d) and d[0] >= 100:
                votes[d[0]] += 1
                last[d[0]] = max(last.get(d[0], 0), ps[-1])
    if not votes:
        return 0, 0
    p = votes.most_common(1)[0][0]
    return p, len(names) - (last[p] + 1)


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
    by_time = "--by-time" in sys.argv
    print(f"# kernel time {tot_ns / steps / 1e6:.3f} ms per step; kernels sorted by {'time' if by_time else 'launch count'}")
    for n, (k_, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1 if by_time else 0])[:70]:
        print(f"{k_ / steps:7.1f}/step {ns / k_ / 1e3:8.2f} us avg {ns / steps / 1e6:7.3f} ms/step  {n[:110]}")


if __name__ == "__main__":
    main()
