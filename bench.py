#!/usr/bin/env python
"""Benchmark of the Dino U-Net training hot path on MI355X (BASELINE.json metric):
2D slices/s for a full training step (forward + DC/CE loss + backward + grad-clip 12 + Nesterov SGD) of `dinounet_l`,
512x512, bf16, batch 8 per GPU, synthetic data, random-init weights; 1..8 GPUs data-parallel (one process per GPU,
RCCL all-reduce of the trainable gradients overlapped with backward).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Rank 0 prints ONE JSON line.  `roofline` is measured live with HIP events around every launch of the dominant kernel
inside the timed region; `cpu_baseline` times the CPU oracle (port of the reference algorithm) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("DINOUNET_ALLOW_RANDOM_BACKBONE", "1")   # synthetic benchmark: random-init weights of the named architecture

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MFMA_BF16_PEAK_TFLOPS = 2500.0   # /opt/skills/guides/MI355X_MICROARCH.md: dense bf16 MFMA peak
HBM_PEAK_GBS = 8000.0
MFMA_BF16_MEASURED_TFLOPS = 2382.0   # same guide: micro-benchmark ceiling of v_mfma_f32_32x32x16_bf16
HBM_MEASURED_GBS = 6290.0            # same guide: float4 copy
PMC_ROUNDS = ("r06", "r05", "r04", "r03", "r02")           # profiles/<round>_pmc_*: counter summaries, newest first; only one measured on the built kernel sources is used


def _pmc_file(stem):
    """newest profiles/<round>_<stem> whose header carries the digest of the kernel sources this library was built from, else the newest
    existing one (the caller then reports the mismatch), else None"""
    from dinounet_amd import _build
    here = os.path.dirname(os.path.abspath(__file__))
    digest = _build._digest()
    found = [os.path.join(here, "profiles", f"{r}_{stem}") for r in PMC_ROUNDS]
    found = [f for f in found if os.path.exists(f)]
    for f in found:
        if any(l.startswith("# csrc-digest") and digest in l for l in open(f).read().splitlines()[:6]):
            return f
    return found[0] if found else None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="dinounet_l")
    ap.add_argument("--batch", type=int, default=8, help="slices per GPU")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--graph", default=os.environ.get("DINOUNET_BENCH_GRAPH", "auto"), choices=["auto", "on", "off"],
                    help="capture the train step into a hipGraph (auto = on)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--torch-sgd", action="store_true", help="torch clip_grad_norm_ + torch.optim.SGD instead of the fused HIP clip+SGD (same arithmetic)")
    return ap.parse_args()


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    force_pg = os.environ.get("DINOUNET_FORCE_REDUCER") == "1" and "RANK" in os.environ   # single-rank RCCL smoke test
    if world > 1 or force_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)   # "nccl" is RCCL on ROCm

    from dinounet_amd.plans import PLANS_2D
    from dinounet_amd import ops
    from dinounet_amd.network_architecture import DinoUNet
    from dinounet_amd.parallel import GradAllReducer
    from dinounet_amd.training import TrainStep

    torch.manual_seed(1234)
    net = DinoUNet.from_config(PLANS_2D, 3, 2, dinov3_pretrained_path=None, dinov3_model_name=a.model, precision=a.precision)
    net = net.to(dev).train()
    params = [p for p in net.parameters() if p.requires_grad]
    reducer = GradAllReducer(net, world) if (world > 1 or force_pg) else None
    if a.torch_sgd:
        opt = torch.optim.SGD(params, lr=1e-3, momentum=0.99, nesterov=True, weight_decay=3e-5)   # nnUNetTrainer.py:486, DT:1024
    else:
        from dinounet_amd.optim import FusedClipSGD                                                # same update + clip 12 (TRN:922), csrc/optim.hip
        opt = FusedClipSGD(params, lr=1e-3, momentum=0.99, nesterov=True, weight_decay=3e-5, max_norm=12.0)

    g = torch.Generator(device="cpu").manual_seed(100 + rank)
    x = torch.randn(a.batch, 3, a.size, a.size, generator=g).to(dev)
    tgt = torch.randint(0, 2, (a.batch, 1, a.size, a.size), generator=g).to(dev)
    use_graph = a.graph != "off"
    ts = TrainStep(net, opt, params, x.shape, tgt.shape, dev, reducer=reducer, graph=use_graph, warmup=max(2, min(a.warmup, 3)))
    ts(x, tgt)

    def step():
        return ts()

    # W untimed warm-up steps (the first ones run eagerly; with --graph the capture happens inside the warm-up)
    for _ in range(max(a.warmup, (4 if use_graph else 1)) - 1):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    graph_used = ts.graph is not None          # False when capture was off or fell back to eager steps
    # roofline leg: the same step run eagerly with a HIP-event pair around every launch of the GEMM / attention kernels
    prof = None
    if not a.no_roofline or reducer is not None:
        # every rank runs these eager steps (they contain the gradient / SyncBN / Dice collectives); only rank 0 times its launches.
        # With a reducer they run even under --no-roofline: the `comm` record (exposed all-reduce time, bucket timeline) comes from them
        ts.use_graph = False
        ts()
        torch.cuda.synchronize()
        if rank == 0 and not a.no_roofline:
            ops.PROFILE = ops.KernelProfile()
        if reducer is not None:
            ts.comm_events = []
            reducer.timeline = []
            step_ev = []
            ops.SMALL_COLLECTIVES = []       # SyncBN / Dice all-reduces of these eager steps, bracketed by compute-stream events
        for _ in range(2):
            if reducer is not None:
                e = torch.cuda.Event(enable_timing=True); e.record(); step_ev.append(e)
            ts()
        torch.cuda.synchronize()
    prof = ops.PROFILE
    ops.PROFILE = None
    comm = None
    if reducer is not None:
        # what the N > 1 runs exchange per step and how much of it is exposed: bucket sizes, collective count, the time the (eager)
        # step spends in reducer.finish() after backward = all-reduce time NOT hidden behind the backward pass + the bucket divides, and
        # per bucket (last eager step) when its gradients were gathered, when its all-reduce started and ended, in ms from the step's start
        nb = [int(f.numel()) for f in reducer.flat]
        exposed = [e0.elapsed_time(e1) for e0, e1 in (ts.comm_events or [])]
        tl = []
        nbk = len(nb)
        for bi, ev in (reducer.timeline or [])[-nbk:]:
            tl.append({"bucket": bi, "elems": nb[bi], "ready_ms": round(step_ev[-1].elapsed_time(ev[0]), 3),
                       "allreduce_start_ms": round(step_ev[-1].elapsed_time(ev[1]), 3), "allreduce_done_ms": round(step_ev[-1].elapsed_time(ev[2]), 3)})
        step_end = None
        if ts.comm_events:
            step_end = round(step_ev[-1].elapsed_time(ts.comm_events[-1][1]), 3)
        small = ops.SMALL_COLLECTIVES or []
        ops.SMALL_COLLECTIVES = None
        small_cnt, small_ms = {}, 0.0
        for what, e0, e1 in small:
            small_cnt[what] = small_cnt.get(what, 0) + 1
            small_ms += e0.elapsed_time(e1)
        comm = {"capture": ts.capture_mode,     # whole_step (RCCL kernels inside the hipGraph) | segments(n) (DINOUNET_COMM_OUTSIDE_GRAPH=1 or
                                                # the fallback when the whole-step capture fails: collectives issued between n graphs) | eager
                "gradient_buckets_elems": nb, "gradient_bytes_per_step": 4 * sum(nb), "gradient_allreduces_per_step": len(nb),
                # measured on the 2 eager steps: how many latency-bound all-reduces (SyncBatchNorm statistics, batch-Dice sums) a step issues
                # and how long the compute stream spends in them (they sit on the critical path; absent at world size 1)
                "small_collectives_per_step": {k: v // 2 for k, v in small_cnt.items()},
                "small_collectives_ms_per_step": round(small_ms / 2, 3) if small else None,
                "exposed_after_backward_ms": round(sum(exposed) / max(len(exposed), 1), 3) if exposed else None,
                "bucket_timeline": tl, "gradients_installed_ms": step_end,
                "note": "eager steps; reducer.finish() = wait for the side-stream all-reduces + divide by world size; timeline in ms from the "
                        "start of the last eager step (compute-stream events for `ready`; `allreduce_done_ms` is recorded on the side stream "
                        "BEHIND Work.wait(), i.e. after the collective itself has finished on RCCL's stream)"}
        ts.comm_events = None
        reducer.timeline = None
    if world > 1:
        t = torch.tensor([dt], device=dev)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        per_rank_ms = [round(float(v.item()) / a.steps * 1e3, 3) for v in every]      # spread of the ranks' own clocks around the same K steps
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        if comm is not None:
            comm["ms_per_step_by_rank"] = per_rank_ms
    ms = dt / a.steps * 1e3
    value = a.batch * world / (dt / a.steps)

    # BASELINE.json's metric for the default workload; other --model / --size / --precision runs name what they measured
    metric = f"2D slices/sec training, {a.model} {a.size}x{a.size} {a.precision}"
    out = {"metric": metric, "value": round(value, 3), "unit": "slices/s",
           "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": a.precision, "data": "synthetic",
           "config": {"workload": f"{a.model} train step (fwd+loss+bwd+clip+SGD), {a.size}x{a.size}x3 slices, batch {a.batch}/GPU, "
                                  f"frozen ViT + adapter + FAPM + U-Net decoder, random-init weights",
                      "global_batch": a.batch * world, "parallelism": f"dp{world}"},
           "final_loss": round(float(loss.item()), 5), "hipgraph": bool(graph_used)}
    if comm is not None:
        out["comm"] = comm
    # the only trustworthy round-over-round delta is an interleaved same-box A/B of the two trees (box-to-box spread of one tree: ~15 %);
    # its factor is committed beside the raw runs and quoted here so it need not be dug out
    try:
        ab = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "ab_vs_prev_round.json")))
        out["ab_vs_prev_round"], out["ab_source"] = ab["factor"], ab["source"]
    except Exception:  # noqa: BLE001
        pass
    if rank == 0:
        if prof is not None:
            try:
                out["roofline"], out["kernel_breakdown"] = prof.roofline(MFMA_BF16_PEAK_TFLOPS, HBM_PEAK_GBS, 2)
                r = out["roofline"]
                r["timing"] = ("HIP events around EAGER launches of the same step (a kernel cannot be timed alone inside the replayed graph); the "
                               "rocprofv3 kernel trace of the replayed graph is committed under profiles/ (steady-state table): for the GEMM "
                               "kernels eager and replayed durations agree to ~1 %")
                r["targets"] = north_star_targets(prof, 2)
                meas = MFMA_BF16_MEASURED_TFLOPS if r.get("bound") == "mfma" else HBM_MEASURED_GBS
                r["measured_peak"] = meas                      # what a micro-benchmark reaches on this chip (same unit as peak)
                r["frac_of_measured_peak"] = round(r["achieved"] / meas, 4)
                pmc_traffic(r)
                pmc_cycles(r)
            except Exception as e:  # noqa: BLE001
                out["roofline"] = {"error": repr(e)}
        if not a.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(net, a)
            except Exception as e:  # noqa: BLE001
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1 or force_pg:
        # Ordered teardown.  The captured step holds RCCL kernel nodes and the reducer a side stream with hooks into the module: if the
        # interpreter tears these down in arbitrary order at exit (process group first, or never destroyed), ProcessGroupNCCL's watchdog
        # can abort after the JSON line is out.  So: quiesce, drop the graph, detach the reducer, then destroy the process group.
        torch.cuda.synchronize()
        dist.barrier()
        if ts.graph is not None:
            ts.graph.reset()
            ts.graph = None
        if reducer is not None:
            reducer.remove()
        ts.reducer = None
        del reducer
        torch.cuda.synchronize()
        dist.destroy_process_group()


def north_star_targets(prof, steps=1):
    """BASELINE.json north_star targets next to what this run measured (same eager HIP-event timing as `roofline`):
    ViT-L attention >= 40 % of the dense bf16 MFMA peak, decoder 3x3 convolutions >= 60 % of peak HBM bandwidth (algorithmic bytes)."""
    agg = {}
    for name, e0, e1, fl, nb in prof.rec:
        key = name.split("<")[0].split(" ")[0]
        a = agg.setdefault(key, [0.0, 0.0, 0.0])
        a[0] += e0.elapsed_time(e1) * 1e-3
        a[1] += fl
        a[2] += nb
    out = {}
    # every target line names where its number comes from: "north_star" = BASELINE.json, "builder" = a goal this repo set itself
    for ak in ("attn_fwd_w64_kernel", "attn_fwd_kernel"):
        if ak in agg and agg[ak][0] > 0:
            t, fl, _ = agg[ak]
            out["attention_mfma_frac"] = {"measured": round(fl / t / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4), "target": 0.40, "kernel": ak,
                                          "source": "north_star"}
            break
    # decoder 3x3 convolutions, named by the kernel that ran: strip = streaming kernel (32 / 64 channels, W % 128 == 0), halo = LDS-tiled
    # kernel (64 + 64 concat and whatever the strip kernel declines), c128 = LDS-tiled kernel with 128 output channels (above the ridge:
    # reported against the MFMA roof).  `decoder_conv_hbm_frac` keeps its round-4 meaning (all layers with <= 64 output channels, both
    # kernels); `decoder_conv_all_hbm_frac` is the round 1-3 / BASELINE definition (EVERY decoder 3x3 convolution, 128-channel ones included).
    fam = {k: agg[k] for k in ("conv3x3_strip_kernel", "conv3x3_halo_kernel", "conv3x3_halo_c128_kernel") if k in agg and agg[k][0] > 0}
    le64 = [fam[k] for k in ("conv3x3_strip_kernel", "conv3x3_halo_kernel") if k in fam]
    if le64:
        t, nb = sum(v[0] for v in le64), sum(v[2] for v in le64)
        out["decoder_conv_hbm_frac"] = {"measured": round(nb / t / 1e9 / HBM_PEAK_GBS, 4), "target": 0.60, "source": "north_star",
                                        "kernel": "conv3x3_strip_kernel + conv3x3_halo_kernel, Cout <= 64", "ms_per_step": round(t * 1e3 / steps, 3)}
        for k, label in (("conv3x3_strip_kernel", "decoder_conv_strip_hbm_frac"), ("conv3x3_halo_kernel", "decoder_conv_halo_hbm_frac")):
            if k in fam:
                t1, _, nb1 = fam[k]
                out[label] = {"measured": round(nb1 / t1 / 1e9 / HBM_PEAK_GBS, 4), "target": 0.60, "source": "north_star (per kernel)",
                              "kernel": k + ", Cout <= 64", "ms_per_step": round(t1 * 1e3 / steps, 3)}
    if fam:
        t, nb = sum(v[0] for v in fam.values()), sum(v[2] for v in fam.values())
        out["decoder_conv_all_hbm_frac"] = {"measured": round(nb / t / 1e9 / HBM_PEAK_GBS, 4), "target": 0.60, "source": "north_star",
                                            "kernel": "every 3x3 stride-1 convolution of the step (rounds 1-3 definition: 128-channel layers included)",
                                            "ms_per_step": round(t * 1e3 / steps, 3)}
    if "conv3x3_halo_c128_kernel" in fam:   # 128 output channels: above the ridge -> MFMA roof
        t, fl, _ = fam["conv3x3_halo_c128_kernel"]
        out["decoder_conv_mfma_frac"] = {"measured": round(fl / t / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4), "target": 0.35, "source": "builder",
                                         "kernel": "conv3x3_halo_kernel, Cout = 128", "ms_per_step": round(t * 1e3 / steps, 3)}
    for key, name, tgt in (("msda_fwd", "msda_fwd_hbm_frac", 0.35), ("msda_bwd", "msda_bwd_hbm_frac", 0.25)):
        if key in agg and agg[key][0] > 0:                                        # the gather path of MSDeformAttn (algorithmic bytes)
            t, _, nb = agg[key]
            out[name] = {"measured": round(nb / t / 1e9 / HBM_PEAK_GBS, 4), "target": tgt, "source": "builder", "gbs": round(nb / t / 1e9, 1),
                         "ms_per_step": round(t * 1e3 / steps, 3)}
    out["scaling_8gpu"] = {"measured": None, "target": 6.5, "source": "north_star", "note": "the driver's SCALE run measures it"}
    return out


def _pmc_match(tag, trace_name):
    """does a rocprofv3 kernel name belong to the profile family `tag` (ops.KernelProfile)?  The persistent kernel's instantiations with a
    residual (template argument RES != 0, incl. the ConvTranspose + skip form) are the `+res` family: HBM-bound products that must not be
    averaged into the plain family's traffic."""
    import re
    key = tag.split("<")[0]
    if key not in trace_name:
        return False
    if key == "gemm_nt_pp_kernel":
        m = re.search(r"gemm_nt_pp_kernel<\s*(\d+),\s*(true|false),\s*(\d+)", trace_name)
        if m:
            return (m.group(3) != "0") == ("+res" in tag)
    return True


def pmc_traffic(roof):
    """roofline.traffic = memory-side bytes per launch of the dominant kernel, from rocprofv3 PMC passes of this same command
    (tools/pmc_traffic.py: `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` in separate runs of `bench.py --graph off`; counters cannot be
    collected from inside the process being timed).  The summaries record the digest of the kernel sources they were measured on
    (dinounet_amd/_build.py: sha256 over csrc/, the C-ABI header and the compiler flags): a summary taken on other kernels is ignored and
    traffic stays null.  FETCH_SIZE / WRITE_SIZE are kilobytes; on gfx950 FETCH_SIZE counts 128-byte requests at 64 B, hence the factor 2
    (MI355X_MICROARCH.md, HBM section)."""
    from dinounet_amd import _build
    here = os.path.dirname(os.path.abspath(__file__))
    key = roof.get("kernel", "").split("<")[0]
    digest = _build._digest()
    tot, src = {}, []
    for name, mult in (("fetch", 2.0), ("write", 1.0)):
        path = _pmc_file(f"pmc_{name}_size_eager.txt")
        if not key or path is None:
            return
        lines = open(path).read().splitlines()
        if not any(l.startswith("# csrc-digest") and digest in l for l in lines[:6]):
            roof["traffic_note"] = f"profiles/{os.path.basename(path)} was measured on other kernel sources (digest mismatch): ignored"
            return
        calls, kb = 0, 0.0
        for line in lines:
            f = line.split(None, 2)
            if len(f) == 3 and f[0].isdigit() and _pmc_match(roof.get("kernel", ""), f[2]):
                calls += int(f[0]); kb += int(f[0]) * float(f[1])
        if not calls:
            return
        tot[name] = kb / calls * 1024.0 * mult
        src.append(os.path.basename(path))
    roof["traffic"] = round(tot["fetch"] + tot["write"])
    roof["traffic_unit"] = "bytes/launch"
    roof["traffic_source"] = f"profiles/{{{','.join(src)}}} (separate rocprofv3 --pmc passes of bench.py --graph off on csrc digest {digest[:12]}; FETCH_SIZE x2)"


def pmc_cycles(roof):
    """roofline.pmc_cycles = where the dominant kernel's cycles go, from the SQ-counter passes of this same command (tools/pmc_kernels.py ->
    profiles/<round>_pmc_sq_cycles_eager.txt, digest-checked like the traffic files): share of the launch the matrix pipe is busy (the PMC
    counterpart of `frac`), and the shares of the resident waves' lifetime spent parked (s_waitcnt / barrier), issue-stalled and issuing."""
    from dinounet_amd import _build
    path = _pmc_file("pmc_sq_cycles_eager.txt")
    key = roof.get("kernel", "").split("<")[0]
    if not key or path is None:
        return
    lines = open(path).read().splitlines()
    if not any(l.startswith("# csrc-digest") and _build._digest() in l for l in lines[:3]):
        return
    n, acc = 0, [0.0] * 5
    for line in lines:
        if line.startswith("#") or not line.strip():
            if line.startswith("# raw means"):
                break
            continue
        f = line.replace("|", " ").split()
        if len(f) >= 12 and f[0].isdigit() and _pmc_match(roof.get("kernel", ""), line):
            try:
                vals = [float(f[i]) for i in (2, 3, 5, 6, 7)]          # parked, stall, issue, mfma, valu
            except ValueError:
                continue
            w = int(f[0])
            n += w
            acc = [a_ + w * v for a_, v in zip(acc, vals)]
    if n:
        parked, stall, issue, mfma, valu = [round(v / n, 3) for v in acc]
        roof["pmc_cycles"] = {"matrix_pipe_busy": mfma, "valu_issue": valu, "waves_parked": parked, "waves_issue_stalled": stall,
                              "waves_issuing": issue, "source": f"profiles/{os.path.basename(path)} (eager launches, launch-weighted over the "
                                                                 "kernel's instantiations)"}


def cpu_baseline(net, a):
    """The CPU oracle (port of the reference's algorithm, oracle/dinounet_oracle.py) timed on the host cores on a
    bounded sample of the same workload: one full train step (forward + loss + backward) of the same model at the same
    resolution with batch 1 (the reference has no CPU MSDA backward; autograd through its grid_sample formula is what
    ops/test.py gradchecks against)."""
    from oracle import dinounet_oracle as O
    sd = {k: v.detach().float().cpu() for k, v in net.state_dict().items()}
    named = dict(net.named_parameters())
    for k in sd:
        if k in named and named[k].requires_grad:
            sd[k] = sd[k].clone().requires_grad_(True)
    for k in list(sd):
        if k.startswith("decoder.encoder."):
            sd[k] = sd[k[len("decoder."):]]
    B = 1
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, 3, a.size, a.size, generator=g)
    tgt = torch.randint(0, 2, (B, 1, a.size, a.size), generator=g)
    cores = torch.get_num_threads()
    t0 = time.perf_counter()
    y = O.dinounet_forward(x, sd, a.model, training=True)
    loss = O.dc_and_ce_loss(y, tgt)
    leaves = [v for k, v in sd.items() if v.requires_grad and not k.startswith("decoder.encoder.") and ".all_modules." not in k]
    torch.autograd.grad(loss, leaves, allow_unused=True)
    dt = time.perf_counter() - t0
    return {"value": round(B / dt, 4), "unit": "slices/s", "cores": cores, "kind": "port",
            "sample": f"1 train step (fwd+loss+bwd, fp32) of {a.model} on {B} slice {a.size}x{a.size}, {dt:.1f} s on {cores} threads "
                      f"({os.cpu_count()} logical CPUs)"}


if __name__ == "__main__":
    main()
