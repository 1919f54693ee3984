#!/bin/bash
# The N > 1 step executed on ONE MI355X: a one-rank RCCL group, the gradient-bucket reducer and -- forced by
# DINOUNET_FORCE_SMALL_COLLECTIVES=1 -- every SyncBatchNorm / batch-Dice all-reduce.  Headline workload (dinounet_l 512^2 batch 8), the three
# forms TrainStep can land in: whole-step capture, segmented capture (collectives from the host between the graphs), eager steps.
# usage: bash tools/forced_collectives_rates.sh [steps] > profiles/rNN_forced_collectives_rates.txt
S=${1:-20}
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 DINOUNET_FORCE_REDUCER=1
pick='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); c=d.get("comm") or {}
        print(json.dumps({"value": d["value"], "ms_per_step": d["ms_per_step"], "hipgraph": d["hipgraph"], "capture": c.get("capture"), "small_collectives_per_step": c.get("small_collectives_per_step"), "small_collectives_ms_per_step": c.get("small_collectives_ms_per_step"), "gradient_allreduces_per_step": c.get("gradient_allreduces_per_step"), "exposed_after_backward_ms": c.get("exposed_after_backward_ms")}))'
echo "# plain single-GPU step (no process group)"
env -u RANK -u DINOUNET_FORCE_REDUCER python bench.py --steps $S --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "$pick"
echo "# reducer only (round-5 configuration: gradient buckets, no small collectives)"
MASTER_PORT=29611 python bench.py --steps $S --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "$pick"
echo "# + every SyncBN / Dice collective: whole-step capture"
MASTER_PORT=29612 DINOUNET_FORCE_SMALL_COLLECTIVES=1 python bench.py --steps $S --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "$pick"
echo "# + every SyncBN / Dice collective: segmented capture (DINOUNET_COMM_OUTSIDE_GRAPH=1)"
MASTER_PORT=29613 DINOUNET_FORCE_SMALL_COLLECTIVES=1 DINOUNET_COMM_OUTSIDE_GRAPH=1 python bench.py --steps $S --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "$pick"
echo "# + every SyncBN / Dice collective: eager steps (--graph off)"
MASTER_PORT=29614 DINOUNET_FORCE_SMALL_COLLECTIVES=1 python bench.py --steps $S --warmup 5 --no-cpu-baseline --no-roofline --graph off 2>/dev/null | python -c "$pick"
echo "# soak: the forced whole-step capture, 10 consecutive processes (rc of each)"
for i in 1 2 3 4 5 6 7 8 9 10; do MASTER_PORT=$((29620+i)) DINOUNET_FORCE_SMALL_COLLECTIVES=1 python bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-roofline >/dev/null 2>&1; echo -n "rc=$? "; done; echo
