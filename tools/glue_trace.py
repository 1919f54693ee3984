#!/usr/bin/env python
"""GPU tuning aid: which Python call sites launch the small torch kernels (copies, fills, adds, casts) that remain around the
HIP ops in one eager train step of the bench workload.  A TorchDispatchMode records every watched aten call with the innermost
frames inside this repo (backward Functions included) and prints the call sites ordered by launch count.
usage: python tools/glue_trace.py [--model dinounet_l] [--batch 8] [--top 60]"""
import argparse
import collections
import os
os.environ.setdefault("DINOUNET_ALLOW_RANDOM_BACKBONE", "1")
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

WATCH = ("aten::copy_", "aten::fill_", "aten::zero_", "aten::add", "aten::add_", "aten::mul", "aten::mul_", "aten::cat", "aten::_to_copy",
         "aten::clone", "aten::contiguous", "aten::sum", "aten::div", "aten::sub", "aten::neg", "aten::zeros", "aten::zeros_like",
         "aten::index", "aten::slice_backward", "aten::select_backward", "aten::masked_fill_", "aten::where", "aten::stack")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="dinounet_l")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--top", type=int, default=60)
    ap.add_argument("--trainstep", action="store_true", help="trace the step bench.py runs (training.TrainStep eager step with FusedClipSGD) and EVERY aten op that is not a pure view, backward included (autograd multithreading off so the dispatch mode sees it)")
    ap.add_argument("--profiler", action="store_true", help="torch.profiler view instead: every watched op with input shapes, split by thread (the autograd engine's own copies / zero fills are invisible to the dispatch-mode tracer)")
    a = ap.parse_args()
    from dinounet_amd.plans import PLANS_2D
    from dinounet_amd.network_architecture import DinoUNet
    from dinounet_amd.training import dc_and_ce_loss
    dev = torch.device("cuda", 0)
    torch.manual_seed(1234)
    net = DinoUNet.from_config(PLANS_2D, 3, 2, dinov3_pretrained_path=None, dinov3_model_name=a.model, precision="bf16").to(dev).train()
    params = [p for p in net.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=1e-3, momentum=0.99, nesterov=True, weight_decay=3e-5)
    g = torch.Generator(device="cpu").manual_seed(100)
    x = torch.randn(a.batch, 3, a.size, a.size, generator=g).to(dev)
    tgt = torch.randint(0, 2, (a.batch, 1, a.size, a.size), generator=g).to(dev)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = dc_and_ce_loss(net(x), tgt)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 12.0)
        opt.step()

    if a.trainstep:
        from dinounet_amd.training import TrainStep
        from dinounet_amd.optim import FusedClipSGD
        opt2 = FusedClipSGD(params, lr=1e-3, momentum=0.99, nesterov=True, weight_decay=3e-5, max_norm=12.0)
        ts = TrainStep(net, opt2, params, x.shape, tgt.shape, dev, graph=False)
        ts(x, tgt)
        step = lambda: ts()
        torch.autograd.set_multithreading_enabled(False)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    if a.profiler:
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
            step()
            torch.cuda.synchronize()
        main_thread = None
        rows = collections.Counter()
        for ev in prof.events():
            if main_thread is None and ev.name.startswith("aten::"):
                main_thread = ev.thread
            if ev.name in ("aten::copy_", "aten::fill_", "aten::add", "aten::add_", "aten::mul", "aten::cat", "aten::index", "aten::div", "aten::neg", "aten::sum"):
                shp = str([tuple(x) for x in (ev.input_shapes or []) if x])[:70]
                rows[(ev.name, "fwd" if ev.thread == main_thread else "bwd", shp)] += 1
        print(f"# {sum(rows.values())} launches-ish in one eager step (leaf aten ops only)")
        for (name, th, shp), n in rows.most_common(a.top):
            print(f"{n:5d}  {th}  {name:14s} {shp}")
        return
    import traceback
    from torch.utils._python_dispatch import TorchDispatchMode
    sites = collections.Counter()

    class Tracer(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = "aten::" + func.__name__.split(".")[0]
            views = ("view", "reshape", "_unsafe_view", "as_strided", "t", "transpose", "permute", "expand", "slice", "select", "unsqueeze", "squeeze",
                     "detach", "alias", "empty", "empty_like", "empty_strided", "split", "split_with_sizes", "unbind", "narrow", "flatten", "unflatten",
                     "_local_scalar_dense", "is_same_size", "sym_size", "stride", "size", "numel", "lift_fresh", "view_as", "chunk", "_reshape_alias",
                     "new_empty", "new_empty_strided", "resize_", "set_", "is_pinned", "record_stream", "contiguous")
            watched = (func.__name__.split(".")[0] not in views) if a.trainstep else (name in WATCH or name in ("aten::empty_strided", "aten::bernoulli_", "aten::_foreach_add_", "aten::_foreach_mul_"))
            if watched:
                frames = [f for f in traceback.extract_stack() if ("dinounet_amd" in f.filename or "tools/" in f.filename) and "glue_trace" not in f.filename]
                where = " < ".join(f"{os.path.relpath(f.filename, ROOT)}:{f.lineno}" for f in frames[-3:][::-1]) or "(autograd engine / optimizer)"
                shp = ""
                for x_ in args:
                    if torch.is_tensor(x_):
                        shp = str(tuple(x_.shape)); break
                    if isinstance(x_, (list, tuple)) and x_ and isinstance(x_[0], int):
                        shp = str(tuple(x_)); break
                sites[(name, where, shp)] += 1
            return func(*args, **(kwargs or {}))

    with Tracer():
        step()
    torch.cuda.synchronize()
    print(f"# {sum(sites.values())} watched aten calls in one eager step")
    for (name, where, shp), n in sites.most_common(a.top):
        print(f"{n:5d}  {name:18s} {shp:28s} {where}")


if __name__ == "__main__":
    main()
