#!/usr/bin/env python
"""Kernels of a rocprofv3 rocpd SQLite database (--kernel-trace) sorted by LAUNCH COUNT: launches per step, average duration, ms per step.
A launch costs ~5 us inside the replayed train-step graph whatever it does, so this is the to-do list for launch merging.
usage: python tools/rocpd_counts.py <results.db> <steps in the trace>"""
import sqlite3,sys,re
c=sqlite3.connect(sys.argv[1]); steps=int(sys.argv[2])
rows=c.execute("select name, count(*), sum(end-start) from kernels group by name order by count(*) desc").fetchall()
tot=sum(r[1] for r in rows)
print(f"# {tot} dispatches = {tot/steps:.0f} per step; kernels sorted by launch count")
for n,k,ns in rows[:45]:
    print(f"{k/steps:7.1f}/step {ns/k/1e3:8.2f} us avg {ns/steps/1e6:7.3f} ms/step  {n[:110]}")
