#!/usr/bin/env python
"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel: calls, mean counter value per launch (and x64 B for the TCC
request counters).  usage: python tools/pmc_summary.py <dir with *_counter_collection.csv> [top_n]"""
import csv
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return name if len(name) <= 110 else name[:107] + "..."


def main():
    d = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    agg = {}
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = (row["Kernel_Name"], row["Counter_Name"])
                a = agg.setdefault(k, [0, 0.0])
                a[0] += 1
                a[1] += float(row["Counter_Value"])
    counters = sorted({c for _, c in agg})
    kernels = {}
    for (k, c), (n, v) in agg.items():
        kernels.setdefault(k, {})[c] = (n, v)
    print(f"# {len(files)} file(s); counters: {counters}")
    order = sorted(kernels.items(), key=lambda kv: -max(v[1] for v in kv[1].values()))[:top]
    print(f"{'calls':>7} " + " ".join(f"{c + '/launch':>22}" for c in counters) + "  kernel")
    for k, cs in order:
        n = max(v[0] for v in cs.values())
        print(f"{n:7d} " + " ".join(f"{(cs[c][1] / cs[c][0]) if c in cs else float('nan'):22.1f}" for c in counters) + "  " + short(k))


if __name__ == "__main__":
    main()
