#!/usr/bin/env python
"""The grouped weight-gradient launch of one dinounet_l train step (ops.WgradQueue -> du_gemm_tn_group), job by job: M (outputs), N (inputs x
taps), K (pixels / tokens), the 256 x 256 tiles and K-tile pairs of each job, and -- mirroring the library's scheduler (gemm_p8.hip
tn_group_launch) -- how the 256 workgroups of each launch are dealt out: units, K-tile pairs per unit (max / mean = the tail a launch waits
for), padded-tile fraction.  VERDICT r5 next #7: "print the per-job table, find the partial-tile / tail-round loss".
usage: python tools/wgrad_jobs.py [--json out.json]"""
import json
import os
import sys
os.environ.setdefault("DINOUNET_ALLOW_RANDOM_BACKBONE", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def schedule(jobs, target_units=256, min_pairs=8, max_jobs=32):
    """mirror of tn_group_launch: consecutive jobs while their tiles fit one round; then split the job with the longest units"""
    launches, i0 = [], 0
    while i0 < len(jobs):
        tiles, pairs, splits, n, units = [], [], [], 0, 0
        while i0 + n < len(jobs) and n < max_jobs:
            g, M, N, K = jobs[i0 + n][:4]
            t = ((M + 255) // 256) * ((N + 255) // 256)
            if n > 0 and units + t > target_units:
                break
            tiles.append(t); pairs.append(K // 128); splits.append(1); units += t; n += 1
        while True:
            best, ln = -1, 0.0
            for k in range(n):
                if units + tiles[k] > target_units or pairs[k] // (splits[k] + 1) < min_pairs:
                    continue
                l = pairs[k] / splits[k]
                if l > ln:
                    ln, best = l, k
            if best < 0:
                break
            splits[best] += 1; units += tiles[best]
        launches.append([(jobs[i0 + k], tiles[k], pairs[k], splits[k]) for k in range(n)])
        i0 += n
    return launches


def report(trace):
    tot_fl = 0.0
    for li, jobs in enumerate(trace):
        by_kind = {0: [], 2: [], 3: []}
        for j in jobs:
            by_kind[j[0]].append(j)
        for kind, js in by_kind.items():
            if not js:
                continue
            for la in schedule(js):
                units = sum(t * s for _, t, _, s in la)
                per_unit = [-(-p // s) for _, t, p, s in la for _ in range(t * s)]
                fl = sum(2.0 * j[1] * j[2] * j[3] for j, _, _, _ in la)
                useful = sum(j[1] * j[2] * p for j, t, p, s in la)                 # live tile area x K pairs
                padded = sum(t * 65536 * p for j, t, p, s in la)
                tot_fl += fl
                print(f"flush {li} kind {kind}: {len(la):2d} jobs {units:3d} units | K-tile pairs per unit max {max(per_unit):4d} mean {sum(per_unit) / len(per_unit):7.1f} "
                      f"(over 256 CUs {sum(per_unit) / 256:7.1f}) | live tile area {useful / padded:.2f} | {fl / 1e9:7.1f} GF | "
                      f"ideal at 1.25 PF {fl / 1.25e15 * 1e6:6.1f} us, max-unit bound {max(per_unit) * 2 * 2 * 256 * 256 * 64 / (1.25e15 / 256) * 1e6:6.1f} us")
                for j, t, p, s in la:
                    print(f"      M {j[1]:5d} N {j[2]:5d} K {j[3]:7d}  tiles {t:3d} pairs {p:5d} splits {s:2d} -> {-(-p // s):4d} pairs/unit{'  per-sample scale' if j[4] else ''}")
    print(f"total {tot_fl / 1e12:.3f} TF")


def main():
    import torch
    from dinounet_amd import ops
    from dinounet_amd.plans import PLANS_2D
    from dinounet_amd.network_architecture import DinoUNet
    from dinounet_amd.training import dc_and_ce_loss
    dev = torch.device("cuda", 0)
    torch.manual_seed(1234)
    net = DinoUNet.from_config(PLANS_2D, 3, 2, dinov3_pretrained_path=None, dinov3_model_name="dinounet_l", precision="bf16").to(dev).train()
    g = torch.Generator(device="cpu").manual_seed(100)
    x = torch.randn(8, 3, 512, 512, generator=g).to(dev)
    tgt = torch.randint(0, 2, (8, 1, 512, 512), generator=g).to(dev)
    for it in range(2):
        ops.WGRAD.trace = [] if it == 1 else None
        net.zero_grad(set_to_none=True)
        dc_and_ce_loss(net(x), tgt).backward()
        ops.WGRAD.flush()
    torch.cuda.synchronize()
    trace = ops.WGRAD.trace
    ops.WGRAD.trace = None
    if "--json" in sys.argv:
        json.dump(trace, open(sys.argv[sys.argv.index("--json") + 1], "w"))
    report(trace)


if __name__ == "__main__":
    if "--from" in sys.argv:
        report(json.load(open(sys.argv[sys.argv.index("--from") + 1])))
    else:
        main()
