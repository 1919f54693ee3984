#!/usr/bin/env python
"""Reproducer: `Work.wait()` on an async RCCL collective issued INSIDE a hipGraph capture makes
`hipStreamEndCapture` fault (torch 2.10.0+rocm7.0, one-rank `nccl` group, capture_error_mode="thread_local").  Without the wait the same
capture ends and replays fine -- which is why parallel.GradAllReducer records its bucket "done" events behind Work.wait() on EAGER steps
only and orders captured steps with stream waits.  An 8-GPU debugging session that adds a wait() inside the captured step will hit this.

usage: python tools/repro_work_wait_in_capture.py          # runs both variants in child processes and prints their exit codes
       python tools/repro_work_wait_in_capture.py child {wait|nowait}"""
import os
import subprocess
import sys


def child(variant):
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29655")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", device_id=dev)
    x = torch.ones(1 << 20, device=dev)
    side = torch.cuda.Stream()
    dist.all_reduce(x)                       # communicator set up outside any capture
    torch.cuda.synchronize()
    import time
    time.sleep(0.35)                         # let the watchdog retire the warm-up work (training.TrainStep does the same)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        y = x * 2.0
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            w = dist.all_reduce(y, async_op=True)
            if variant == "wait":
                w.wait()                     # <- the call that makes capture_end() fault
        torch.cuda.current_stream().wait_stream(side)
        z = y + 1.0
    print("capture ended", flush=True)
    g.replay()
    torch.cuda.synchronize()
    print("replayed, z[0] =", float(z[0]), flush=True)
    g.reset()
    dist.destroy_process_group()


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "child":
        child(sys.argv[2])
        sys.exit(0)
    for i, variant in enumerate(("nowait", "wait")):
        env = dict(os.environ, MASTER_PORT=str(29655 + i))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", variant], env=env, capture_output=True, text=True, timeout=300)
        tail = [l for l in (r.stdout + r.stderr).splitlines() if l.strip() and "amdgpu.ids" not in l][-4:]
        print(f"{variant:7s} rc {r.returncode}  | " + " | ".join(t[:160] for t in tail), flush=True)
