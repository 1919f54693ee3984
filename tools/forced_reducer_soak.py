#!/usr/bin/env python
"""Soak test of the single-rank RCCL path: N consecutive runs of bench.py with DINOUNET_FORCE_REDUCER=1 (live `nccl`
process group, bucketed all-reduces on the side stream inside the hipGraph-captured step, ordered teardown) -- every one must exit 0
without a retry.  With --outside as the first extra argument the runs set DINOUNET_COMM_OUTSIDE_GRAPH=1 (the step captured in segments,
collectives issued between the graphs) and check `comm.capture`.
usage: python tools/forced_reducer_soak.py [runs] [--outside] [extra bench.py arguments ...]"""
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
outside = len(sys.argv) > 2 and sys.argv[2] == "--outside"
if outside:
    del sys.argv[2]
extra = sys.argv[2:] or ["--model", "dinounet_s", "--size", "256", "--batch", "2", "--steps", "6", "--warmup", "4"]
bad = 0
t00 = time.time()
for i in range(n):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, DINOUNET_FORCE_REDUCER="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if outside:
        env["DINOUNET_COMM_OUTSIDE_GRAPH"] = "1"
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-roofline", *extra], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    ok = r.returncode == 0 and line and json.loads(line[-1]).get("hipgraph") and json.loads(line[-1])["comm"]["exposed_after_backward_ms"] is not None
    ok = ok and json.loads(line[-1])["comm"]["capture"].startswith("segments" if outside else "whole_step")
    bad += 0 if ok else 1
    print(f"run {i + 1:2d}/{n}: rc {r.returncode} {'ok' if ok else 'FAILED'} ({time.time() - t0:.1f} s)" + ("" if ok else "\n" + r.stderr[-3000:]), flush=True)
print(f"{n - bad}/{n} runs clean in {time.time() - t00:.0f} s ({' '.join(extra)}; capture {'in segments, collectives outside the graphs' if outside else 'of the whole step'})")
sys.exit(1 if bad else 0)
