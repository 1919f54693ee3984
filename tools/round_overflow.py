#!/usr/bin/env python
"""Launches of a rocprofv3 --kernel-trace database whose workgroups overflow a whole number of ROUNDS on the chip by a little: the last few
workgroups then run a round of their own (a full latency chain for a sliver of the work) -- how the 40 ragged rows of the ViT products
 and the forty extra waves of its LayerNorms showed up.  Per distinct (kernel, grid, block) of the steady-state step: launches per
step, average duration, workgroups, resident workgroups per CU (from threads, LDS and registers where the database has them), rounds =
workgroups / (256 CUs x resident), and the time share of launches with 1.0 < rounds < 1.15 or 2.0 < rounds < 2.1.
usage: python tools/round_overflow.py <results.db> [CUs=256]"""
import collections
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    cus = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    cols = [r[1] for r in c.execute("pragma table_info(kernels)").fetchall()]
    print("# columns of `kernels`:", ", ".join(cols))
    pick = lambda *names: next((n for n in names if n in cols), None)
    gx, gy, gz = pick("grid_x", "grid_size_x"), pick("grid_y", "grid_size_y"), pick("grid_z", "grid_size_z")
    wx, wy, wz = pick("workgroup_x", "workgroup_size_x"), pick("workgroup_y", "workgroup_size_y"), pick("workgroup_z", "workgroup_size_z")
    lds, vg, ag = pick("lds_size", "lds_block_size", "group_segment_size"), pick("vgpr_count", "arch_vgpr_count"), pick("accum_vgpr_count")
    if not (gx and wx):
        print("no grid / workgroup columns in this database")
        return
    sel = ["name", "start", "end", gx, gy or "1", gz or "1", wx, wy or "1", wz or "1", lds or "0", vg or "0", ag or "0"]
    rows = c.execute(f"select {', '.join(sel)} from kernels order by start").fetchall()
    rows = rows[len(rows) // 2:]                     # the second half of the trace: replayed steps only
    agg = collections.OrderedDict()
    for nm, s, e, a, b, cc, x, y, z, l, v, acc in rows:
        threads = x * y * z
        grid = a * b * cc                            # rocprofv3 reports the grid in work-items
        wgs = grid // max(threads, 1)
        key = (nm[:90], wgs, threads, l, v, acc)
        t = agg.setdefault(key, [0, 0.0])
        t[0] += 1
        t[1] += (e - s) / 1e3
    out = []
    for (nm, wgs, threads, l, v, acc), (n, us) in agg.items():
        waves = (threads + 63) // 64
        res = 32 // max(waves, 1)                   # wave slots: 8 per SIMD, 32 per CU
        if l:
            res = min(res, max(1, 163840 // max(l, 1)))
        regs = (v or 0) + (acc or 0)
        if regs:
            per_simd = max(1, 512 // max(regs, 1))
            res = min(res, max(1, per_simd * 4 // max(waves, 1)))
        rounds = wgs / (cus * max(res, 1))
        out.append((us, n, us / n, wgs, threads, l, regs, res, rounds, nm))
    out.sort(reverse=True)
    tot = sum(o[0] for o in out)
    print(f"# {'us total':>10} {'n':>5} {'us avg':>8} {'WGs':>7} {'thr':>4} {'LDS':>7} {'regs':>4} {'res/CU':>6} {'rounds':>7}  kernel")
    flagged = 0.0
    for us, n, avg, wgs, threads, l, regs, res, rounds, nm in out[:120]:
        frac = rounds - int(rounds)
        flag = "  <== overflow" if (rounds > 1.0 and rounds < 4.0 and 0.0 < frac < 0.13) else ""
        if flag:
            flagged += us
        print(f"  {us:10.1f} {n:5d} {avg:8.2f} {wgs:7d} {threads:4d} {l:7d} {regs:4d} {res:6d} {rounds:7.3f}  {nm}{flag}")
    print(f"# flagged launches hold {flagged / max(tot, 1e-9):.1%} of the traced kernel time")


if __name__ == "__main__":
    main()
