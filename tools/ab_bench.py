#!/usr/bin/env python
"""Interleaved same-box A/B of whole trees and / or option sets on the headline bench ("every claimed gain is an interleaved
same-box A/B of the round-4 .so against the new one, bench.py --steps 30, >= 3 alternations, committed under profiles/").

    python tools/ab_bench.py --rounds 3 --steps 30 --arm r04=build/r04 --arm r05=. --arm r05_1chain=.,DINOUNET_VIT_CHAINS=1

An arm = name=<tree>[,ENV=value...]: `python <tree>/bench.py --steps S --warmup W --no-cpu-baseline --no-roofline` is run with cwd = the
tree (so each tree loads its OWN dinounet_amd package and libdinounet_hip.so) and the given environment additions.  The arms are run
round-robin (A B C A B C ...), after one discarded warm-up run of the first arm (clocks, page cache).  Output: one line per run and a
per-arm summary (median / min / max slices/s and ms per step, ratio of medians against the first arm)."""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arm", action="append", required=True, help="name=tree[,ENV=value...]")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--extra", default="", help="extra bench.py arguments for every arm, e.g. '--model dinounet_s --batch 16'")
    ap.add_argument("--timeout", type=int, default=600)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    arms = []
    for spec in a.arm:
        name, rest = spec.split("=", 1)
        parts = rest.split(",")
        tree = os.path.abspath(os.path.join(root, parts[0]))
        env = dict(p.split("=", 1) for p in parts[1:])
        assert os.path.exists(os.path.join(tree, "bench.py")), tree
        arms.append((name, tree, env))
    lines = []

    def say(s):
        print(s, flush=True)
        lines.append(s)

    def run(name, tree, env):
        cmd = [sys.executable, os.path.join(tree, "bench.py"), "--steps", str(a.steps), "--warmup", str(a.warmup), "--no-cpu-baseline", "--no-roofline"]
        cmd += a.extra.split()
        e = dict(os.environ, **env)
        e.pop("PYTHONPATH", None)
        t0 = time.time()
        r = subprocess.run(cmd, cwd=tree, env=e, capture_output=True, text=True, timeout=a.timeout)
        js = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not js:
            say(f"# {name}: rc {r.returncode}\n{r.stdout[-1500:]}\n{r.stderr[-3000:]}")
            return None
        d = json.loads(js[-1])
        return d["value"], d["ms_per_step"], time.time() - t0, d.get("hipgraph")

    say(f"# interleaved A/B, bench.py --steps {a.steps} --warmup {a.warmup} {a.extra} (no roofline / cpu legs), {a.rounds} rounds, arms: "
        + "; ".join(f"{n} = {os.path.relpath(t, root)} {e if e else ''}" for n, t, e in arms))
    say(f"# box: {os.uname().nodename}; started {time.strftime('%Y-%m-%d %H:%M:%S')}")
    w = run(*arms[0])
    say(f"# discarded warm-up run of {arms[0][0]}: {w}")
    res = {n: [] for n, _, _ in arms}
    for rd in range(a.rounds):
        for name, tree, env in arms:
            r = run(name, tree, env)
            if r is None:
                continue
            res[name].append(r)
            say(f"round {rd}  {name:16s} {r[0]:9.2f} slices/s  {r[1]:8.3f} ms/step  (hipgraph {r[3]}, run {r[2]:.0f} s)")
    base = None
    say("# summary (median [min .. max])")
    for name, _, _ in arms:
        v = [x[0] for x in res[name]]
        m = [x[1] for x in res[name]]
        if not v:
            say(f"{name:16s} no successful run")
            continue
        med = statistics.median(v)
        if base is None:
            base = med
        say(f"{name:16s} {med:9.2f} slices/s [{min(v):.2f} .. {max(v):.2f}]   {statistics.median(m):8.3f} ms/step [{min(m):.3f} .. {max(m):.3f}]   "
            f"x{med / base:.4f} vs {arms[0][0]}")
    if a.out:
        with open(a.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
