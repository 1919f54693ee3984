#!/usr/bin/env python
"""Measure roofline.traffic for bench.py: two rocprofv3 PMC passes (FETCH_SIZE, then WRITE_SIZE: they do not fit one pass) of
`bench.py --graph off --steps 2 --warmup 1`, summarised per kernel into profiles/r06_pmc_{fetch,write}_size_eager.txt with the digest of
the kernel sources they were measured on (bench.py ignores a summary whose digest differs from the library it runs).
Run ON THE GPU BOX:  python tools/pmc_traffic.py   (writes under gpurun_out/, copy the two summaries into profiles/)."""
import glob
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dinounet_amd import _build  # noqa: E402


def summarise(db, counter, out, header):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), avg(counter_value) from pmc_events where counter_name = ? group by name "
                     "order by sum(counter_value) desc", (counter,)).fetchall()
    with open(out, "w") as f:
        f.write(header)
        f.write(f"# {'calls':>7} {'mean ' + counter + ' per launch (KB)':>36}  kernel\n")
        for name, n, v in rows[:60]:
            f.write(f"{n:9d} {v:36.1f}  {name[:160]}\n")


def main():
    outdir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(outdir, exist_ok=True)
    digest = _build._digest()
    sha = os.environ.get("GIT_SHA", "unknown (no .git on the GPU box; see the commit that adds this file)")
    env = dict(os.environ, TMPDIR="/tmp")
    for name, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        d = os.path.join("/tmp", f"pmc_{name}")
        cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "-d", d, "-o", "p", "--", sys.executable, os.path.join(ROOT, "bench.py"),
               "--graph", "off", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-roofline"]
        subprocess.run(cmd, check=True, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        db = glob.glob(os.path.join(d, "**", "*results.db"), recursive=True)[0]
        header = (f"# csrc-digest {digest}\n# git {sha}\n# rocprofv3 --pmc {counter} --kernel-trace -- python bench.py --graph off --steps 2 "
                  f"--warmup 1 --no-cpu-baseline --no-roofline   (mean counter value per launch, KB; FETCH_SIZE counts 128-B requests at "
                  f"64 B on gfx950: x2)\n")
        summarise(db, counter, os.path.join(outdir, f"r06_pmc_{name}_size_eager.txt"), header)
        print(open(os.path.join(outdir, f"r06_pmc_{name}_size_eager.txt")).read()[:1500])


if __name__ == "__main__":
    main()
