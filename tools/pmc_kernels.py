#!/usr/bin/env python
"""Where do the wave cycles of each kernel of a dinounet_l train step go?  Two rocprofv3 PMC passes (SQ has 8 counter slots per pass,
MI355X_MICROARCH.md "rocprofv3 PMC slots") over `bench.py --graph off --steps 2 --warmup 1`, summarised per kernel:

  pass A  SQ_WAVE_CYCLES  SQ_BUSY_CYCLES  SQ_WAIT_ANY  SQ_WAIT_INST_ANY  SQ_WAIT_INST_LDS  SQ_ACTIVE_INST_ANY
          SQ_VALU_MFMA_BUSY_CYCLES  SQ_LDS_BANK_CONFLICT  + GRBM_GUI_ACTIVE
  pass B  SQ_LDS_IDX_ACTIVE  SQ_LDS_UNALIGNED_STALL  SQ_ACTIVE_INST_VALU  SQ_ACTIVE_INST_LDS  SQ_ACTIVE_INST_VMEM  SQ_ACTIVE_INST_SCA
          SQ_INSTS_VALU  SQ_WAVES  + GRBM_GUI_ACTIVE

Counters the installed rocprofv3 does not list (`rocprofv3 -L`) are dropped from a pass instead of failing it.  Derived columns (see render()): the wave view -- parked / stalled / issuing shares of the waves' lifetime -- and the SIMD view -- duty of
the matrix pipe, the VALU, LDS and VMEM issue ports over the launch -- plus LDS bank-conflict cycles per LDS-array cycle.
Run ON THE GPU BOX:  python tools/pmc_kernels.py   -> gpurun_out/r06_pmc_sq_cycles_eager.txt (copy into profiles/);
re-render an earlier output without a GPU:  python tools/pmc_kernels.py --from-raw <file>."""
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


def render(data, used, n_inst, path):
    """data: kernel -> counter -> (rows, mean per counter INSTANCE).  SQ counters come as one row per (XCD, shader engine) = n_inst rows per
    dispatch (32 on MI355X: SQ_WAVES x 32 = the launch's waves), each covering 1024 / n_inst SIMDs; GRBM_GUI_ACTIVE is the launch's length
    in shader cycles in every row.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles."""
    simd = N_SIMD / n_inst

    def g(d, k):
        return d[k][1] if k in d else None

    def ratio(a, b):
        return f"{a / b:6.3f}" if a is not None and b else "     -"

    rows = []
    for name, d in data.items():
        wc, gui = g(d, "SQ_WAVE_CYCLES"), g(d, "GRBM_GUI_ACTIVE")
        n = d.get("SQ_WAVE_CYCLES", (0, 0))[0] // n_inst
        rows.append((gui * n if gui else 0.0, name, n, d, wc, gui))
    rows.sort(key=lambda r: -r[0])
    with open(path, "w") as f:
        f.write(f"# csrc-digest {_build._digest()}\n")
        for tag, cs in used.items():
            f.write(f"# pass {tag}: rocprofv3 --pmc {' '.join(cs)} --kernel-trace -- python bench.py --graph off --steps 2 --warmup 1 "
                    f"--no-cpu-baseline --no-roofline\n")
        f.write("# dinounet_l 512x512 bf16 batch 8, three eager steps; means over all launches of a kernel name; kernels sorted by launches x "
                "GRBM_GUI_ACTIVE.\n"
                f"# SQ counters arrive per (XCD, shader engine): {n_inst} rows per dispatch, {simd:.0f} SIMDs each (SQ_WAVES x {n_inst} = waves of the "
                "launch).\n"
                "# wave view (fractions of SQ_WAVE_CYCLES, i.e. of the resident waves' lifetime):\n"
                "#   parked = WAIT_ANY (s_waitcnt / s_barrier), stall = WAIT_INST_ANY (issue stalls: MFMA dependency, busy pipe; lds = its "
                "WAIT_INST_LDS part), issue = ACTIVE_INST_ANY\n"
                "# SIMD view (fractions of the launch, GUI_ACTIVE x SIMDs; the launch length includes the eager dispatch gap):\n"
                "#   mfma = VALU_MFMA_BUSY_CYCLES, valu = 4 x ACTIVE_INST_VALU (MFMA issue included), ldsi = 4 x ACTIVE_INST_LDS, vmem = 4 x "
                "ACTIVE_INST_VMEM\n"
                "# conflict = LDS_BANK_CONFLICT / LDS_IDX_ACTIVE (extra cycles per LDS-array cycle)\n")
        f.write(f"# {'launches':>8} {'gui_active':>11} {'parked':>6} {'stall':>6} {'lds':>6} {'issue':>6} | {'mfma':>6} {'valu':>6} {'ldsi':>6} "
                f"{'vmem':>6} | {'conflict':>8}  kernel\n")
        for _, name, n, d, wc, gui in rows[:80]:
            den = gui * simd if gui else None

            def duty(k, mul):
                v = g(d, k)
                return ratio(v * mul if v is not None else None, den)
            f.write(f"{n:10d} {gui or 0:11.0f} {ratio(g(d, 'SQ_WAIT_ANY'), wc)} {ratio(g(d, 'SQ_WAIT_INST_ANY'), wc)} "
                    f"{ratio(g(d, 'SQ_WAIT_INST_LDS'), wc)} {ratio(g(d, 'SQ_ACTIVE_INST_ANY'), wc)} | "
                    f"{duty('SQ_VALU_MFMA_BUSY_CYCLES', 1)} {duty('SQ_ACTIVE_INST_VALU', 4)} {duty('SQ_ACTIVE_INST_LDS', 4)} "
                    f"{duty('SQ_ACTIVE_INST_VMEM', 4)} | {ratio(g(d, 'SQ_LDS_BANK_CONFLICT'), g(d, 'SQ_LDS_IDX_ACTIVE')):>8}  {short(name)}\n")
        f.write("\n# raw means per counter instance\n")
        for _, name, n, d, wc, gui in rows[:80]:
            f.write(f"{short(name)}\n")
            for cn in sorted(d):
                f.write(f"    {cn:28s} {d[cn][1]:16.1f}   ({d[cn][0]} rows)\n")


def from_raw(path):
    """Rebuild (data, used) from the raw section of an earlier output (re-render without the GPU)."""
    data, used, cur = {}, {}, None
    raw = False
    for line in open(path):
        m = re.match(r"# pass (\w): rocprofv3 --pmc (.*?) --kernel-trace", line)
        if m:
            used[m.group(1)] = m.group(2).split()
        elif line.startswith("# raw means"):
            raw = True
        elif raw and line.startswith("    "):
            cn, v, n = re.match(r"\s+(\S+)\s+([0-9.]+)\s+\((\d+) ", line).groups()
            data[cur][cn] = (int(n), float(v))
        elif raw and line.strip():
            cur = line.rstrip("\n")
            data[cur] = {}
    return data, used


def main():
    outdir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(outdir, exist_ok=True)
    path = os.path.join(outdir, "r06_pmc_sq_cycles_eager.txt")
    if len(sys.argv) > 2 and sys.argv[1] == "--from-raw":
        data, used = from_raw(sys.argv[2])
        render(data, used, 32, path)
        print(open(path).read()[:5000])
        return
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
    # rows per dispatch of an SQ counter: the launch's waves / the per-row SQ_WAVES of a kernel whose grid we know is not available here,
    # so take the smallest row count ratio SQ : GRBM seen x the GRBM rows per dispatch (1 or 4 by rocprofv3 version) -> fall back to 32
    render(data, used, 32, path)
    print(open(path).read()[:5000])


if __name__ == "__main__":
    main()
