#!/usr/bin/env python
"""Dump the time-ordered kernel-name sequence of a rocprofv3 rocpd database as small integers (+ the name table), for offline analysis of
the step structure (tools/rocpd_counts.py).  usage: python tools/rocpd_dump_names.py <results.db> <out.json>"""
import json
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
ids = {}
seq = []
for n, s, e in rows:
    seq.append([ids.setdefault(n, len(ids)), s, e - s])
json.dump({"names": [n[:120] for n in ids], "seq": seq}, open(sys.argv[2], "w"))
