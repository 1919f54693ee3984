"""GPU parity tests (-m gpu), end to end: the product DinoUNet (HIP kernels through the C-ABI) against the committed
outputs of the REFERENCE'S OWN modules (tests/golden/*, generated in the build container by oracle/make_golden.py) with
identical synthetic weights (oracle/weights.py) and inputs.

fp32 kernel mode: logits within 1e-3 relative (north_star), argmax masks identical except where the reference's own
top-2 margin is below the tolerance.  bf16 throughput mode: deviation reported and bounded separately."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import dinounet_oracle as O
from oracle import weights
from oracle.refshim import PLANS_2D

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    return g, json.loads(str(g["meta"]))


def _build(model, num_classes, precision, deep_supervision=False):
    from dinounet_amd.network_architecture import DinoUNet
    plans = PLANS_2D if not deep_supervision else dict(PLANS_2D, architecture=dict(PLANS_2D["architecture"], deep_supervision=True))
    net = DinoUNet.from_config(plans, 3, num_classes, dinov3_pretrained_path=None, dinov3_model_name=model, precision=precision)
    ks = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    net.load_state_dict(weights.make_state_dict(ks, seed=0), strict=True)
    return net.cuda()


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    assert torch.isfinite(a).all()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-20))


def argmax_check(y, ref, tol):
    """masks must agree wherever the reference's top-2 margin exceeds tol * max|logit|"""
    y, ref = y.detach().float().cpu(), ref.float()
    mism = y.argmax(1) != ref.argmax(1)
    top2 = ref.topk(2, dim=1).values
    margin = top2[:, 0] - top2[:, 1]
    bad = mism & (margin > tol * ref.abs().max())
    return int(mism.sum()), int(bad.sum())


CASES = ["dinounet_s_64_eval", "dinounet_s_96x64_eval", "dinounet_s_64_c1_eval", "dinounet_s_64_k4_eval", "dinounet_b_64_eval",
         "dinounet_l_64_eval", "dinounet_s_512_eval", "dinounet_b_512_eval", "dinounet_l_512_eval"]


@pytest.mark.parametrize("name", CASES)
def test_eval_logits_fp32_vs_reference(name):
    g, meta = _load(name)
    net = _build(meta["model"], meta["num_classes"], "fp32").eval()
    x = weights.make_input(meta["B"], meta["C"], meta["H"], meta["W"], seed=0).cuda()
    with torch.no_grad():
        y = net(x)
    ref = torch.from_numpy(g["logits"])
    e = rel(y, ref)
    mism, bad = argmax_check(y, ref, 1e-3)
    npix = ref.argmax(1).numel()
    print(f"[{name}] fp32 HIP vs reference: rel err {e:.2e}, argmax mismatches {mism} of {npix} pixels (outside tie band: {bad})")
    assert e < 1e-3
    assert bad == 0
    # "argmax masks bit-exact" is tested as: identical wherever the reference's own top-2 margin exceeds 1e-3 * max|logit| (a pixel inside
    # that band flips under ANY fp32 reordering of the reference itself).  The band must stay a measure-zero curiosity: at most 1 pixel in
    # 10 000 (and never more than 20) may differ inside it
    assert mism <= min(20, max(1, npix // 10000)), (mism, npix)


@pytest.mark.parametrize("name", ["dinounet_s_64_eval", "dinounet_l_64_eval", "dinounet_s_512_eval"])
def test_eval_logits_bf16_mode(name):
    g, meta = _load(name)
    net = _build(meta["model"], meta["num_classes"], "bf16").eval()
    x = weights.make_input(meta["B"], meta["C"], meta["H"], meta["W"], seed=0).cuda()
    with torch.no_grad():
        y = net(x)
    ref = torch.from_numpy(g["logits"])
    e = rel(y, ref)
    agree = float((y.float().cpu().argmax(1) == ref.argmax(1)).float().mean())
    print(f"[{name}] bf16 HIP vs reference fp32: rel err {e:.2e}, argmax agreement {agree:.4f}")
    assert e < 0.15 and agree > 0.97


def test_stage_taps_fp32():
    """Per-stage tensors of the reference (ViT taps, c after each interaction, f1..f4, FAPM, skips): localises a wrong kernel."""
    g, meta = _load("dinounet_s_64_eval")
    net = _build("dinounet_s", 2, "fp32").eval()
    taps = {}
    ad = net.encoder.dinov3_adapter
    for i, blk in enumerate(ad.interactions):
        blk.register_forward_hook(lambda m, a, out, i=i: taps.__setitem__(f"c{i + 1}", out))   # __setitem__ returns None
    # NB: a forward hook must return None, otherwise its return value replaces the module output
    def h_feats(m, a, out):
        for j, k in enumerate("1234"):
            taps[f"feats{j}"] = out[k].permute(0, 3, 1, 2)

    def h_list(prefix):
        def h(m, a, out):
            for j, o in enumerate(out):
                taps[f"{prefix}{j}"] = o.permute(0, 3, 1, 2)
        return h

    ad.register_forward_hook(h_feats)
    net.encoder.fapm.register_forward_hook(h_list("fapm"))
    net.encoder.register_forward_hook(h_list("skips"))
    orig = ad.backbone.get_intermediate_layers

    def gil(*a, **k):
        r = orig(*a, **k)
        for j, t in enumerate(r):
            taps[f"vit{j}"] = t[0]
        return r

    ad.backbone.get_intermediate_layers = gil
    x = weights.make_input(2, 3, 64, 64, seed=0).cuda()
    with torch.no_grad():
        y = net(x)
    errs = {k: rel(v, torch.from_numpy(g[k])) for k, v in taps.items()}
    errs["logits"] = rel(y, torch.from_numpy(g["logits"]))
    for k in sorted(errs):
        print(f"  tap {k:8s} rel err {errs[k]:.2e}")
    assert len(errs) == 21
    assert max(errs.values()) < 1e-3, errs


def test_train_step_fp32_vs_reference():
    """train(): batch-statistics BN, DC+CE loss, backward through every HIP backward kernel; per-parameter gradient norms and
    the small gradients in full vs the reference's (drop-path / RoPE jitter disabled on both sides)."""
    from dinounet_amd.dinov3.adapter import DropPath
    g, meta = _load("dinounet_s_64_train")
    net = _build("dinounet_s", 2, "fp32").train()
    for m in net.modules():
        if isinstance(m, DropPath):
            m.drop_prob = 0.0
    net.encoder.dinov3_adapter.backbone.rope_embed.rescale_coords = None
    x = weights.make_input(2, 3, 64, 64, seed=1).cuda()
    tgt = weights.make_target(2, 64, 64, 2, seed=1).cuda()
    y = net(x)
    loss = O.dc_and_ce_loss(y, tgt)
    loss.backward()
    assert rel(y, torch.from_numpy(g["logits"])) < 1e-3
    assert abs(loss.item() - float(g["loss"])) < 1e-4
    norms = meta["grad_norms"]
    gmax = max(norms.values())
    named = dict(net.named_parameters())
    worst = ("", 0.0)
    for k, n in norms.items():
        assert named[k].grad is not None, k
        got = float(named[k].grad.norm())
        err = abs(got - n) / max(n, 1e-3 * gmax)
        if err > worst[1]:
            worst = (k, err)
    print(f"worst grad-norm deviation {worst[1]:.2e} at {worst[0]}")
    assert worst[1] < 1e-2, worst
    errs = []
    for key in g.files:
        if key.startswith("grad:"):
            k = key[5:]
            ref = torch.from_numpy(g[key])
            if ref.norm() > 1e-3 * gmax:
                errs.append((float((named[k].grad.cpu() - ref).norm() / ref.norm()), float(ref.norm() / gmax), k))
    errs.sort(reverse=True)
    for e, n, k in errs[:6]:
        print(f"  full-gradient rel-L2 {e:.2e}  (|g| / max|g| = {n:.1e})  {k}")
    assert errs[0][0] < 2e-2, errs[0]
    assert sorted(k for k, p in named.items() if p.requires_grad and p.grad is None) == meta["unused"]


def test_deep_supervision_train_step_fp32_vs_reference():
    """UNetDecoder with deep_supervision=True (dinounet_training.py:603-629): one head per decoder stage, list largest first.  The three
    logits tensors, a weighted Dice + CE over them and every gradient norm (small gradients in full) against the reference's own modules
    (tests/golden/dinounet_s_64_ds_train.npz, oracle/make_golden_ds.py); all three seg layers receive gradients."""
    from dinounet_amd.dinov3.adapter import DropPath
    g, meta = _load("dinounet_s_64_ds_train")
    net = _build("dinounet_s", 2, "fp32", deep_supervision=True).train()
    assert net.decoder.deep_supervision is True
    for m in net.modules():
        if isinstance(m, DropPath):
            m.drop_prob = 0.0
    net.encoder.dinov3_adapter.backbone.rope_embed.rescale_coords = None
    x = weights.make_input(2, 3, 64, 64, seed=1).cuda()
    tgt = weights.make_target(2, 64, 64, 2, seed=1).cuda()
    outs = net(x)
    assert isinstance(outs, list) and len(outs) == 3
    loss = 0.0
    for i, (y, w) in enumerate(zip(outs, meta["ds_weights"])):
        assert rel(y, torch.from_numpy(g[f"logits{i}"])) < 1e-3, i
        loss = loss + w * O.dc_and_ce_loss(y, tgt[..., ::2 ** i, ::2 ** i].contiguous())
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 1e-4
    norms = meta["grad_norms"]
    gmax = max(norms.values())
    named = dict(net.named_parameters())
    worst = ("", 0.0)
    for k, n in norms.items():
        assert named[k].grad is not None, k
        err = abs(float(named[k].grad.norm()) - n) / max(n, 1e-3 * gmax)
        if err > worst[1]:
            worst = (k, err)
    print(f"deep supervision: worst grad-norm deviation {worst[1]:.2e} at {worst[0]}")
    assert worst[1] < 1e-2, worst
    errs = []
    for key in g.files:
        if key.startswith("grad:"):
            k = key[5:]
            ref = torch.from_numpy(g[key])
            if ref.norm() > 1e-3 * gmax:
                errs.append((float((named[k].grad.cpu() - ref).norm() / ref.norm()), float(ref.norm() / gmax), k))
    errs.sort(reverse=True)
    # this gradient sits on a discontinuity at 64 x 64 (bilinear samples of 4 x 4 maps on cell borders): a 1e-6 relative perturbation of
    # the input moves the REFERENCE's own gradient of the last extractor's sampling offsets by 2.8 % (meta["self_noise"], measured by
    # oracle/make_golden_ds.py on the reference's modules).  A tensor may deviate by 2e-2 or by 1.5 x the reference's own sensitivity.
    noise = meta["self_noise"]
    for e, n, k in errs[:6]:
        print(f"  full-gradient rel-L2 {e:.2e}  (|g| / max|g| = {n:.1e}, reference self-noise {noise[k]:.2e})  {k}")
    for e, n, k in errs:
        assert e < max(2e-2, 1.5 * noise[k]), (k, e, noise[k])
    for s_ in range(3):
        assert float(named[f"decoder.seg_layers.{s_}.weight"].grad.norm()) > 0
    assert sorted(k for k, p in named.items() if p.requires_grad and p.grad is None) == meta["unused"]


def test_train_step_bf16_runs_and_learns():
    """bf16 throughput mode with drop-path + RoPE jitter on: finite loss/grads, SGD reduces the loss on a fixed batch."""
    net = _build("dinounet_s", 2, "bf16").train()
    torch.manual_seed(0)
    x = weights.make_input(4, 3, 64, 64, seed=3).cuda()
    tgt = weights.make_target(4, 64, 64, 2, seed=3).cuda()
    params = [p for p in net.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, 1e-2, momentum=0.9, nesterov=True)
    losses = []
    for _ in range(8):
        opt.zero_grad(set_to_none=True)
        loss = O.dc_and_ce_loss(net(x), tgt)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 12)
        opt.step()
        losses.append(loss.item())
    assert all(np.isfinite(losses)), losses
    assert losses[-1] < losses[0], losses


def test_bf16_training_trajectory_follows_fp32_mode_512():
    """one more link between the bf16 throughput mode and the reference than single-step gradient bands.  50 optimizer
    steps (the trainer's SGD: Nesterov 0.99, weight decay 3e-5, clip 12, nnUNetTrainer.py:486,922-924) of the bf16 kernels and of the fp32
    parity kernels from IDENTICAL state on the same four 512 x 512 slices (dinounet_s; every DropPath mask and RoPE draw pinned to the same
    values in both runs): the loss curves must stay together (the fp32 mode is what the reference goldens pin), the final logits must give
    the same segmentation, and the Dice of the final masks against the training target must agree."""
    B, steps = 4, 50
    x = weights.make_input(B, 3, 512, 512, seed=6).cuda()
    tgt = weights.make_target(B, 512, 512, 2, seed=6).cuda()
    curves, finals = {}, {}
    for prec in ("fp32", "bf16"):
        net = _build("dinounet_s", 2, prec).train()
        _pin_randomness(net, B)
        params = [p for p in net.parameters() if p.requires_grad]
        opt = torch.optim.SGD(params, 1e-2, momentum=0.99, nesterov=True, weight_decay=3e-5)
        ls = []
        for _ in range(steps):
            opt.zero_grad(set_to_none=True)
            loss = O.dc_and_ce_loss(net(x), tgt)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(params, 12)
            opt.step()
            ls.append(float(loss))
        net.eval()
        with torch.no_grad():
            finals[prec] = net(x).float().cpu()
        curves[prec] = ls
        del net, opt
    c32, c16 = np.array(curves["fp32"]), np.array(curves["bf16"])
    assert np.isfinite(c16).all() and np.isfinite(c32).all()
    drop = c32[0] - c32[-1]
    gap = np.abs(c16 - c32)
    m32, m16 = finals["fp32"].argmax(1), finals["bf16"].argmax(1)
    t = tgt[:, 0].cpu()

    def dice(m):
        tp = float(((m == 1) & (t == 1)).sum()); fp = float(((m == 1) & (t != 1)).sum()); fn = float(((m != 1) & (t == 1)).sum())
        return 2 * tp / max(2 * tp + fp + fn, 1.0)
    agree = float((m32 == m16).float().mean())
    print(f"[trajectory, dinounet_s 4 x 512^2, {steps} steps] loss fp32 {c32[0]:.4f} -> {c32[-1]:.4f}, bf16 {c16[0]:.4f} -> {c16[-1]:.4f}; "
          f"max |gap| {gap.max():.4f} (mean {gap.mean():.4f}) of a {drop:.4f} descent; final masks agree on {agree:.4f} of the pixels; "
          f"Dice vs target fp32 {dice(m32):.4f} bf16 {dice(m16):.4f}")
    assert drop > 0.05, "the fixed batch must be learnable for the comparison to mean anything"
    assert gap.max() < 0.25 * drop and gap.mean() < 0.08 * drop, (gap.max(), gap.mean(), drop)
    assert c16[-1] < c16[0] - 0.6 * drop
    assert agree > 0.97 and abs(dice(m32) - dice(m16)) < 0.03


@pytest.mark.parametrize("model", ["dinounet_s", "dinounet_b", "dinounet_l"])
def test_bf16_mode_vs_fp32_mode_512(model):
    """The throughput (bf16) mode against the parity (fp32) mode of the SAME HIP path at the BASELINE.json shape (512 x 512, batch 2:
    N = 1029 tokens, M = 2058 ragged GEMM rows, Lq = 5376, 512^2 decoder): a wrong edge tile / ragged tail / XCD remap in a
    bf16-only kernel shows up as a block of pixels whose error stands out from the bf16 rounding noise of the rest of the map."""
    x = weights.make_input(2, 3, 512, 512, seed=4).cuda()
    with torch.no_grad():
        y32 = _build(model, 2, "fp32").eval()(x).float()
        y16 = _build(model, 2, "bf16").eval()(x).float()
    assert torch.isfinite(y16).all()
    scale = float(y32.abs().max())
    err = (y16 - y32).abs()
    blk = torch.nn.functional.avg_pool2d(err, 16)                      # mean error of every 16 x 16 pixel block
    glob = float(err.mean())
    mism, bad = argmax_check(y16, y32.cpu(), 0.1)
    print(f"[{model}] bf16 vs fp32 HIP at 512^2: max err {float(err.max()) / scale:.3e}, mean {glob / scale:.3e}, worst 16x16 block "
          f"{float(blk.max()) / scale:.3e} of max|logit|; argmax mismatches {mism} (outside a 10 % margin band: {bad})")
    assert float(err.max()) < 0.12 * scale                               # per pixel
    assert glob < 0.02 * scale
    assert float(blk.max()) < max(6.0 * glob, 0.01 * scale)              # no block of pixels off by more than the noise floor
    assert bad == 0


def _pin_randomness(net, B):
    from dinounet_amd.dinov3.adapter import DropPath
    bb = net.encoder.dinov3_adapter.backbone
    log_scales, masks = weights.pinned_randomness(len(bb.blocks), B, seed=2)
    bb.rope_embed.pinned_log_scales = log_scales
    dps = [m for m in net.modules() if isinstance(m, DropPath)]
    assert len(dps) == len(masks)
    for m, mk in zip(dps, masks):
        m.pinned_mask = mk


@pytest.mark.parametrize("name", ["dinounet_s_64_train_pinned", "dinounet_l_256_train_pinned", "dinounet_s_512_train_pinned",
                                  "dinounet_l_512_train_pinned"])
def test_train_step_pinned_randomness_vs_reference(name):
    """train() with the per-block RoPE rescale draws (LAY/rope_position_encoding.py:93-97) and the DropPath masks (ADP:18-26) pinned to
    the values the reference was given (oracle/make_golden.py: pin_reference_randomness): logits, loss, and EVERY trainable
    gradient -- in full up to `sample` elements, else a fixed `sample`-element subset -- against the reference's."""
    g, meta = _load(name)
    B, H, W = meta["B"], meta["H"], meta["W"]
    net = _build(meta["model"], meta["num_classes"], "fp32").train()
    _pin_randomness(net, B)
    x = weights.make_input(B, 3, H, W, seed=2).cuda()
    tgt = weights.make_target(B, H, W, meta["num_classes"], seed=2).cuda()
    y = net(x)
    loss = O.dc_and_ce_loss(y, tgt)
    loss.backward()
    e = rel(y, torch.from_numpy(g["logits"]))
    print(f"[{name}] logits rel err {e:.2e}, loss {loss.item():.6f} vs {float(g['loss']):.6f}")
    assert e < 1e-3
    assert abs(loss.item() - float(g["loss"])) < 1e-4
    norms = meta["grad_norms"]
    gmax = max(norms.values())
    named = dict(net.named_parameters())
    worst = ("", 0.0)
    n_cmp = 0
    for key in g.files:
        if not key.startswith("grad:"):
            continue
        k = key[5:]
        ref = torch.from_numpy(g[key])
        got = named[k].grad.detach().float().cpu().flatten()
        if got.numel() > meta["sample"]:
            got = got[weights.sample_indices(k, got.numel(), meta["sample"])]
        assert got.shape == ref.shape, k
        # gradients that are analytically zero (biases feeding a norm) carry round-off only: measure against the global scale
        denom = max(float(ref.norm()), 1e-3 * gmax * (ref.numel() / max(named[k].numel(), 1)) ** 0.5)
        err = float((got - ref).norm()) / denom
        n_cmp += 1
        if err > worst[1]:
            worst = (k, err)
    print(f"[{name}] {n_cmp} gradients compared, worst rel-L2 {worst[1]:.2e} at {worst[0]}")
    assert n_cmp == len(norms)
    assert worst[1] < 1e-2, worst       # (measured: 4.5e-3 at dinounet_l 512^2 on spm.stem.4.weight, 6e-3 at dinounet_s 512^2; BASELINE.md quotes <= 1e-2)
    assert sorted(k for k, p in named.items() if p.requires_grad and p.grad is None) == meta["unused"]


# bf16-mode gradients against the fp32-mode gradients of the same HIP network: bound on every tensor's rel-L2 deviation in units of the
# network's own sensitivity (the fp32-mode gradient's response to a bf16-sized perturbation of the INPUT alone), with an absolute floor;
# measured distribution in profiles/r03_bf16_grad_parity.txt
BF16_GRAD_FLOOR, BF16_GRAD_SENS = 5e-2, 4.0


@pytest.mark.parametrize("model,B", [("dinounet_l", 8), ("dinounet_s", 16), ("dinounet_b", 32)])
def test_bf16_train_step_vs_fp32_mode_512(model, B):
    """The configuration bench.py times (BASELINE.json configs 2-4: 512 x 512, batch 8 / 16 / 32, bf16 kernels, train mode with
    DropPath and the RoPE rescale live -- nnUNetTrainer.py:899-929) against the fp32 parity mode of the SAME HIP network with every
    random draw pinned to the same values: forward + Dice/CE + backward once in each mode.  The bf16-only backward kernels that only
    engage at these sizes (grouped / multi-phase TN weight gradients, ConvT gather dgrad / wgrad at 1024 -> 1024, the a_colsum / b_colsum
    bias-gradient owners, DropPath's contraction-row scale and tile skip, M = B * 1029 ragged GEMM rows) are covered here by the model,
    not only by op tests on synthetic shapes.
    Criterion.  Gradients through the deformable attention's sampling locations are ill-conditioned (bilinear kinks, 16 x 16 / 32 x 32
    maps): the fp32 network's OWN gradients move by tens of percent on the low-resolution branch when only its input is rounded to
    bf16 (third run below, `sens`).  A bf16 kernel bug shows up as a tensor that deviates far beyond that sensitivity, as a wrong
    norm ratio, or as a lost direction (cosine): rel-L2 <= max(5e-2, 4 x sens), |g16| / |g32| in [0.7, 1.4], cosine >= 0.85."""
    x = weights.make_input(B, 3, 512, 512, seed=5).cuda()
    tgt = weights.make_target(B, 512, 512, 2, seed=5).cuda()
    res = {}
    for tag, prec, xin in (("fp32", "fp32", x), ("fp32q", "fp32", x.bfloat16().float()), ("bf16", "bf16", x)):
        net = _build(model, 2, prec).train()
        _pin_randomness(net, B)
        y = net(xin)
        loss = O.dc_and_ce_loss(y, tgt)
        loss.backward()
        torch.cuda.synchronize()
        res[tag] = (float(loss.detach()), y.detach().float().cpu(),
                    {k: p.grad.detach().float().cpu() for k, p in net.named_parameters() if p.grad is not None})
        del net, y, loss
        torch.cuda.empty_cache()
    (l32, y32, g32), (_, _, g32q), (l16, y16, g16) = res["fp32"], res["fp32q"], res["bf16"]
    assert np.isfinite(l16) and torch.isfinite(y16).all()
    assert set(g16) == set(g32)
    gmax = max(float(v.norm()) for v in g32.values())
    rows = []
    for k, ref in g32.items():
        got = g16[k]
        assert torch.isfinite(got).all(), k
        # analytically-zero gradients (biases in front of a norm) carry round-off only: measured against the global scale
        denom = max(float(ref.norm()), 1e-3 * gmax)
        err = float((got - ref).norm()) / denom
        sens = float((g32q[k] - ref).norm()) / denom
        big = float(ref.norm()) > 1e-3 * gmax
        ratio = float(got.norm()) / float(ref.norm()) if big else 1.0
        cos = float((got.flatten().double() @ ref.flatten().double()) / (got.norm().double() * ref.norm().double())) if big else 1.0
        rows.append((err, sens, ratio, cos, float(ref.norm()) / gmax, k))
    rows.sort(reverse=True)
    med = float(np.median([r[0] for r in rows]))
    med_s = float(np.median([r[1] for r in rows]))
    print(f"[{model} B{B} 512^2] loss bf16 {l16:.5f} vs fp32 {l32:.5f}; logits rel {rel(y16, y32):.3e}; {len(rows)} gradients: "
          f"median rel-L2 bf16-vs-fp32 {med:.3e}, median fp32 input-rounding sensitivity {med_s:.3e}; worst:")
    for err, sens, ratio, cos, n, k in rows[:10]:
        print(f"    rel-L2 {err:.3e}  sens {sens:.3e}  |g16|/|g32| {ratio:.3f}  cos {cos:.4f}  (|g| / max|g| = {n:.1e})  {k}")
    assert abs(l16 - l32) < 1e-2
    for err, sens, ratio, cos, n, k in rows:
        assert err < max(BF16_GRAD_FLOOR, BF16_GRAD_SENS * sens), (k, err, sens)
        assert 0.7 < ratio < 1.4 and cos > 0.85, (k, ratio, cos)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_vit7b_style_paths_vs_reference(precision):
    """the code paths only the 7B backbone takes -- head dim 128 attention, no qkv bias, the SwiGLU gate (fused into
    the GEMM epilogue in bf16, du_swiglu_pairs in fp32) and the train-mode batch-subset stochastic depth (du_sample_copy gather /
    scatter, scaled residual epilogue) -- on a 7B-style ViT at toy width against the reference's own DinoVisionTransformer
    (tests/golden/vit7b_style_64.npz: eval, and train mode with pinned RoPE draws + pinned subsets)."""
    from dinounet_amd.dinov3.vision_transformer import DinoVisionTransformer
    g = np.load(os.path.join(GOLD, "vit7b_style_64.npz"))
    meta = json.loads(str(g["meta"]))
    cfg, B, depth = meta["cfg"], meta["B"], meta["cfg"]["depth"]
    net = DinoVisionTransformer(**cfg)
    net.load_state_dict(weights.make_state_dict([(k, tuple(s)) for k, s in meta["keys"]], seed=0), strict=True)
    net = net.cuda()
    dt = torch.float32 if precision == "fp32" else torch.bfloat16
    tol = 1e-3 if precision == "fp32" else 4e-2
    x = weights.make_input(B, 3, meta["H"], meta["W"], seed=6).cuda()
    net.eval()
    ev = net.get_intermediate_layers(x, n=list(range(depth)), return_class_token=True, dtype=dt)
    net.train()
    log_scales, _ = weights.pinned_randomness(depth, B, seed=3)
    net.rope_embed.pinned_log_scales = log_scales
    net.pinned_subsets = weights.pinned_subsets(depth, B, cfg["drop_path_rate"], seed=3)
    tr = net.get_intermediate_layers(x, n=list(range(depth)), return_class_token=True, dtype=dt)
    worst = 0.0
    for i in range(depth):
        for mode, outs in (("eval", ev), ("train", tr)):
            worst = max(worst, rel(outs[i][0], torch.from_numpy(g[f"{mode}_patch{i}"])), rel(outs[i][1], torch.from_numpy(g[f"{mode}_cls{i}"])))
    print(f"[vit7b-style {precision}] worst rel err over {depth} blocks x (eval, pinned train) x (patch, cls): {worst:.2e}")
    assert worst < tol
    # un-pinned train mode draws its own subsets on the device: runs, finite, differs from eval
    net.pinned_subsets = None
    net.rope_embed.pinned_log_scales = None
    rnd = net.get_intermediate_layers(x, n=[depth - 1], return_class_token=True, dtype=dt)
    assert torch.isfinite(rnd[0][0].float()).all()


def test_vit7b_width_block_vs_torch_fp32():
    """One block at the true 7B width (D 4096, 32 heads x Dh 128, SwiGLU hidden 8192, no qkv bias) in bf16 against plain torch fp32 of
    the same bf16-rounded weights: the multi-phase GEMMs with the gate epilogue and the Dh-128 attention at production width."""
    from dinounet_amd.dinov3.vision_transformer import DinoVisionTransformer
    cfg = dict(embed_dim=4096, depth=1, num_heads=32, ffn_ratio=3.0, qkv_bias=False, ffn_layer="swiglu64", n_storage_tokens=4,
               mask_k_bias=True)
    net = DinoVisionTransformer(**cfg)
    ks = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    sd = weights.make_state_dict(ks, seed=1)
    net.load_state_dict(sd, strict=True)
    net = net.cuda().eval()
    B, H = 4, 256                                         # 4 x (5 + 256) tokens = 1044 rows: ragged against the 256-row tiles
    x = weights.make_input(B, 3, H, H, seed=8).cuda()
    (patch, cls), = net.get_intermediate_layers(x, n=[0], return_class_token=True, dtype=torch.bfloat16)
    ocfg = dict(embed_dim=4096, depth=1, num_heads=32, ffn="swiglu", qkv_bias=False, interaction_indexes=[0])
    sdq = {k: (v.to(torch.bfloat16).float() if (v.dim() == 2 or k.endswith("patch_embed.proj.weight")) else v) for k, v in sd.items()}
    with torch.no_grad():
        (rp, rc), = O.vit_intermediate(x.cpu(), O.SD(sdq), ocfg)          # the oracle runs on the host
    e = max(rel(patch, rp), rel(cls, rc))
    print(f"[7B-width block bf16 vs torch fp32] rel err {e:.2e}")
    assert e < 4e-2


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_dinounet_7b_width_end_to_end_vs_reference(precision, monkeypatch):
    """`DinoUNet('dinounet_7b')` end to end (hub/backbones.py:452-496 + the adapter / FAPM / decoder at D = 4096: MSDeformAttn head width 128,
    ConvTranspose 4096 -> 4096, FAPM 4096 -> 256) against the reference's own module, backbone cut to depth 4 on both sides so the 0.9 G
    synthetic parameters stay manageable (tests/golden/dinounet_7b_d4_64_eval.npz, oracle/make_golden_7b.py).  fp32 mode: the 1e-3 bar of
    the other end-to-end cases; bf16 mode: the bf16 bounds."""
    from dinounet_amd.dinov3 import vision_transformer as VT
    from dinounet_amd.network_architecture import dinounet as DU
    g, meta = _load("dinounet_7b_d4_64_eval")
    monkeypatch.setitem(VT.VIT_CONFIGS, "dinounet_7b", dict(VT.VIT_CONFIGS["dinounet_7b"], depth=meta["depth"]))
    monkeypatch.setitem(DU.DINOv3_INTERACTION_INDEXES, "dinounet_7b", list(meta["interaction_indexes"]))
    net = DU.DinoUNet.from_config(PLANS_2D, 3, meta["num_classes"], dinov3_pretrained_path=None, dinov3_model_name="dinounet_7b",
                                  precision=precision)
    ks = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    assert len(ks) == meta["n_keys"]
    net.load_state_dict(weights.make_state_dict(ks, seed=0), strict=True)
    net = net.cuda().eval()
    x = weights.make_input(meta["B"], meta["C"], meta["H"], meta["W"], seed=0).cuda()
    with torch.no_grad():
        y = net(x)
    ref = torch.from_numpy(g["logits"])
    e = rel(y, ref)
    agree = float((y.float().cpu().argmax(1) == ref.argmax(1)).float().mean())
    print(f"[dinounet_7b depth {meta['depth']} {precision}] logits rel err {e:.2e}, argmax agreement {agree:.4f}")
    if precision == "fp32":
        mism, bad = argmax_check(y, ref, 1e-3)
        assert e < 1e-3 and bad == 0
    else:
        assert e < 0.15 and agree > 0.97
