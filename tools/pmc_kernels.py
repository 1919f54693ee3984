#!/usr/bin/env python
"""Where do the wave cycles of each kernel of a dinounet_l train step go?  Two rocprofv3 PMC passes (SQ has 8 counter slots per pass,
MI355X_MICROARCH.md "rocprofv3 PMC slots") over `bench.py --graph off --steps 2 --warmup 1`, summarised per kernel:

  pass A  SQ_WAVE_CYCLES  SQ_BUSY_CYCLES  SQ_WAIT_ANY  SQ_WAIT_INST_ANY  SQ_WAIT_INST_LDS  SQ_ACTIVE_INST_ANY
          SQ_VALU_MFMA_BUSY_CYCLES  SQ_LDS_BANK_CONFLICT  + GRBM_GUI_ACTIVE
  pass B  SQ_LDS_IDX_ACTIVE  SQ_LDS_UNALIGNED_STALL  SQ_ACTIVE_INST_VALU  SQ_ACTIVE_INST_LDS  SQ_ACTIVE_INST_VMEM  SQ_ACTIVE_INST_SCA
          SQ_INSTS_VALU  SQ_WAVES  + GRBM_GUI_ACTIVE

Counters the installed rocprofv3 does not list (`rocprofv3 -L`) are dropped from a pass instead of failing it.  Derived columns:
  parked  = WAIT_ANY / WAVE_CYCLES        waves sitting in s_waitcnt / s_barrier
  stall   = WAIT_INST_ANY / WAVE_CYCLES   issue stalls (MFMA dependency, pipe busy); lds = the WAIT_INST_LDS share of it
  issue   = ACTIVE_INST_ANY / WAVE_CYCLES
  mfma    = VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs)      the matrix pipes' duty over the launch (the guide: the counter
            advances 32 per 32x32x16 bf16 MFMA; GUI_ACTIVE is per launch in shader cycles)
  conflict = LDS_BANK_CONFLICT / LDS_IDX_ACTIVE  (pass A / pass B: different launches of the same kernel, means per launch)
Run ON THE GPU BOX:  python tools/pmc_kernels.py   -> gpurun_out/r02_pmc_sq_cycles_eager.txt (copy into profiles/)."""
import glob
import os
import re
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dinounet_amd import _build  # noqa: E402

PASSES = {
    "A": ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY",
          "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_LDS_BANK_CONFLICT", "GRBM_GUI_ACTIVE"],
    "B": ["SQ_LDS_IDX_ACTIVE", "SQ_LDS_UNALIGNED_STALL", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM",
          "SQ_ACTIVE_INST_SCA", "SQ_INSTS_VALU", "SQ_WAVES", "GRBM_GUI_ACTIVE"],
}
N_SIMD = 256 * 4


def available():
    try:
        out = subprocess.run(["rocprofv3", "-L"], capture_output=True, text=True, timeout=120, cwd="/tmp").stdout
    except Exception:                                    # noqa: BLE001  (listing is an optimisation, not a requirement)
        return None
    names = set(re.findall(r"\b((?:SQ|GRBM|TCC|TCP|TA|TD)_[A-Z0-9_]+)\b", out))
    return names or None


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:70]


def collect(tag, counters, env):
    d = os.path.join("/tmp", f"pmc_sq_{tag}")
    cmd = ["rocprofv3", "--pmc", *counters, "--kernel-trace", "-d", d, "-o", "p", "--", sys.executable, os.path.join(ROOT, "bench.py"),
           "--graph", "off", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-roofline"]
    r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        print(f"[pmc_kernels] pass {tag} failed rc={r.returncode}: {r.stderr[-600:]}")
        return {}
    db = glob.glob(os.path.join(d, "**", "*results.db"), recursive=True)[0]
    c = sqlite3.connect(db)
    rows = c.execute("select name, counter_name, count(*), avg(counter_value) from pmc_events group by name, counter_name").fetchall()
    out = {}
    for name, cn, n, v in rows:
        out.setdefault(name, {})[cn] = (n, v)
    return out


def main():
    outdir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(outdir, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    have = available()
    data = {}
    used = {}
    for tag, want in PASSES.items():
        cs = [c for c in want if have is None or c in have]
        used[tag] = cs
        dropped = [c for c in want if c not in cs]
        if dropped:
            print(f"[pmc_kernels] pass {tag}: not listed by rocprofv3 -L, dropped: {dropped}")
        for name, d in collect(tag, cs, env).items():
            data.setdefault(name, {}).update(d)

    def g(d, k):
        return d[k][1] if k in d else None

    def ratio(a, b):
        return f"{a / b:6.3f}" if a is not None and b else "     -"

    rows = []
    for name, d in data.items():
        wc, gui = g(d, "SQ_WAVE_CYCLES"), g(d, "GRBM_GUI_ACTIVE")
        n = d.get("SQ_WAVE_CYCLES", d.get("GRBM_GUI_ACTIVE", (0, 0)))[0]
        rows.append((gui * n if gui else 0.0, name, n, d, wc, gui))
    rows.sort(key=lambda r: -r[0])
    path = os.path.join(outdir, "r02_pmc_sq_cycles_eager.txt")
    with open(path, "w") as f:
        f.write(f"# csrc-digest {_build._digest()}\n")
        for tag, cs in used.items():
            f.write(f"# pass {tag}: rocprofv3 --pmc {' '.join(cs)} --kernel-trace -- python bench.py --graph off --steps 2 --warmup 1 "
                    f"--no-cpu-baseline --no-roofline\n")
        f.write("# means per launch over all launches of a kernel name (dinounet_l 512x512 bf16 batch 8, three eager steps); kernels sorted by "
                "launches x GRBM_GUI_ACTIVE.\n# parked = WAIT_ANY / WAVE_CYCLES, stall = WAIT_INST_ANY / WAVE_CYCLES (lds = WAIT_INST_LDS share), "
                "issue = ACTIVE_INST_ANY / WAVE_CYCLES,\n# mfma = VALU_MFMA_BUSY_CYCLES / (GUI_ACTIVE x 1024 SIMDs), conflict = LDS_BANK_CONFLICT / "
                "LDS_IDX_ACTIVE, valu / ldsi / vmem = ACTIVE_INST_{VALU,LDS,VMEM} / WAVE_CYCLES (pass B over pass A)\n")
        f.write(f"# {'launches':>8} {'gui_active':>11} {'parked':>6} {'stall':>6} {'lds':>6} {'issue':>6} {'mfma':>6} {'conflict':>8} "
                f"{'valu':>6} {'ldsi':>6} {'vmem':>6}  kernel\n")
        for _, name, n, d, wc, gui in rows[:40]:
            mf = g(d, "SQ_VALU_MFMA_BUSY_CYCLES")
            f.write(f"{n:10d} {gui or 0:11.0f} {ratio(g(d, 'SQ_WAIT_ANY'), wc)} {ratio(g(d, 'SQ_WAIT_INST_ANY'), wc)} "
                    f"{ratio(g(d, 'SQ_WAIT_INST_LDS'), wc)} {ratio(g(d, 'SQ_ACTIVE_INST_ANY'), wc)} "
                    f"{ratio(mf, gui * N_SIMD if gui else None)} {ratio(g(d, 'SQ_LDS_BANK_CONFLICT'), g(d, 'SQ_LDS_IDX_ACTIVE')):>8} "
                    f"{ratio(g(d, 'SQ_ACTIVE_INST_VALU'), wc)} {ratio(g(d, 'SQ_ACTIVE_INST_LDS'), wc)} {ratio(g(d, 'SQ_ACTIVE_INST_VMEM'), wc)}  "
                    f"{short(name)}\n")
        f.write("\n# raw means per launch\n")
        for _, name, n, d, wc, gui in rows[:40]:
            f.write(f"{short(name)}\n")
            for cn in sorted(d):
                f.write(f"    {cn:28s} {d[cn][1]:16.1f}   ({d[cn][0]} launches)\n")
    print(open(path).read()[:5000])


if __name__ == "__main__":
    main()
