"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/* by running the REFERENCE'S OWN modules
(imported from /root/reference through oracle/refshim.py) on CPU fp32 with the synthetic weights of
oracle/weights.py, and pins oracle/dinounet_oracle.py against them (asserts agreement).

Run in the build container only:   python -m oracle.make_golden
The GPU box has no /root/reference; it reads the committed fixtures.
"""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import refshim, weights, dinounet_oracle as O  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
TOL = 2e-5


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def build(model_name, num_classes=2, deep_supervision=False):
    net = refshim.build_reference_dinounet(model_name, num_classes=num_classes, deep_supervision=deep_supervision)
    sd0 = net.state_dict()
    ks = [(k, tuple(v.shape)) for k, v in sd0.items()]
    sd = weights.make_state_dict(ks, seed=0)
    net.load_state_dict(sd, strict=True)
    return net, ks, sd


def dump_keys(model_name, ks, sd0):
    with open(os.path.join(GOLD, f"state_dict_{model_name}.json"), "w") as f:
        json.dump({"model": model_name, "n_keys": len(ks),
                   "keys": [[k, list(s), str(sd0[k].dtype).replace("torch.", "")] for k, s in ks]}, f)


def neutralise_train_randomness(net):
    """train() parity: drop-path off (ADP:138) and RoPE rescale off (LAY/rope_position_encoding.py:93-97);
    both are host-RNG driven and reproduced separately by the product (tests share the drawn values)."""
    for m in net.modules():
        if m.__class__.__name__ == "DropPath":
            m.drop_prob = 0.0
    rope = net.encoder.dinov3_adapter.backbone.rope_embed
    rope.rescale_coords = None
    rope.shift_coords = None
    rope.jitter_coords = None


def hook_taps(net):
    """Collect the reference's own per-stage tensors with forward hooks."""
    taps = {}
    ad = net.encoder.dinov3_adapter
    for i, blk in enumerate(ad.interactions):
        blk.register_forward_hook(lambda m, a, out, i=i: taps.__setitem__(f"c{i + 1}", out[1].detach()))
    ad.register_forward_hook(lambda m, a, out: taps.__setitem__("feats", [out[k].detach() for k in "1234"]))
    net.encoder.fapm.register_forward_hook(lambda m, a, out: taps.__setitem__("fapm", [o.detach() for o in out]))
    net.encoder.register_forward_hook(lambda m, a, out: taps.__setitem__("skips", [o.detach() for o in out]))
    orig = ad.backbone.get_intermediate_layers

    def gil(*a, **k):
        r = orig(*a, **k)
        taps["vit"] = [t[0].detach().float() for t in r]
        return r

    ad.backbone.get_intermediate_layers = gil
    return taps


def eval_case(name, model_name, B, C, H, W, num_classes=2, with_taps=False, store_logits="f32"):
    t0 = time.time()
    net, ks, sd = build(model_name, num_classes)
    net.eval()
    x = weights.make_input(B, C, H, W, seed=0)
    taps_ref = hook_taps(net) if with_taps else None
    with torch.no_grad():
        y_ref = net(x)
        taps_or = {} if with_taps else None
        y_or = O.dinounet_forward(x, sd, model_name, training=False, taps=taps_or)
    e = rel(y_or, y_ref)
    print(f"[{name}] oracle-vs-reference logits rel err {e:.2e}  ({time.time() - t0:.1f}s)")
    assert e < TOL, e
    assert torch.equal(y_or.argmax(1), y_ref.argmax(1)) or e < 1e-6
    out = {"logits": y_ref.numpy().astype(np.float32),
           "argmax": np.packbits(y_ref.argmax(1).numpy().astype(np.uint8)) if num_classes == 2 else y_ref.argmax(1).numpy().astype(np.uint8),
           "meta": np.array(json.dumps(dict(model=model_name, B=B, C=C, H=H, W=W, num_classes=num_classes, mode="eval",
                                            oracle_rel_err=e)))}
    if with_taps:
        for k in ("vit", "feats", "fapm", "skips"):
            for i, (a, b) in enumerate(zip(taps_or[k], taps_ref[k])):
                ee = rel(a, b)
                assert ee < TOL, (k, i, ee)
                out[f"{k}{i}"] = b.numpy().astype(np.float32)
        for i in range(1, 5):
            ee = rel(taps_or[f"c{i}"], taps_ref[f"c{i}"])
            assert ee < TOL, (i, ee)
            out[f"c{i}"] = taps_ref[f"c{i}"].numpy().astype(np.float32)
        print(f"[{name}] all per-stage taps agree (<{TOL})")
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    return ks, sd


def train_case(name, model_name, B, H, W, num_classes=2):
    """train() mode forward + DC/CE loss + backward through the reference (MSDA backward = the shimmed
    extension entry point); golden = logits, loss, per-parameter grad L2 norms + a few full small grads."""
    t0 = time.time()
    net, ks, sd = build(model_name, num_classes)
    net.train()
    neutralise_train_randomness(net)
    x = weights.make_input(B, 3, H, W, seed=1)
    tgt = weights.make_target(B, H, W, num_classes, seed=1)
    y_ref = net(x)
    loss_ref = O.dc_and_ce_loss(y_ref, tgt)
    loss_ref.backward()
    named = dict(net.named_parameters())
    g_ref = {k: p.grad for k, p in named.items() if p.requires_grad and p.grad is not None}
    # oracle (restatement) with autograd
    sd2 = {k: (v.clone().requires_grad_(True) if (k in named and named[k].requires_grad) else v) for k, v in sd.items()}
    for k in list(sd2):  # keep aliasing: decoder.encoder.* must be the same leaf as encoder.*
        if k.startswith("decoder.encoder."):
            sd2[k] = sd2[k[len("decoder."):]]
    y_or = O.dinounet_forward(x, sd2, model_name, training=True)
    loss_or = O.dc_and_ce_loss(y_or, tgt)
    leaves = {k: v for k, v in sd2.items() if v.requires_grad and not k.startswith("decoder.encoder.") and ".all_modules." not in k}
    gl = torch.autograd.grad(loss_or, list(leaves.values()), allow_unused=True)
    g_or = {k: g for k, g in zip(leaves, gl) if g is not None}
    e = rel(y_or.detach(), y_ref.detach())
    print(f"[{name}] train-mode logits rel err {e:.2e}; loss ref {loss_ref.item():.6f} oracle {loss_or.item():.6f}")
    assert e < TOL
    worst = 0.0
    norms = {}
    gmax = max(float(g.norm()) for g in g_ref.values())
    for k, g in g_ref.items():
        ck = weights.canonical_key(k)
        norms[k] = float(g.norm())
        # biases feeding a norm layer have analytically-zero gradients (1e-9 round-off): compare against the
        # global gradient scale instead of their own norm
        if ck in g_or:
            worst = max(worst, float((g - g_or[ck]).norm() / max(float(g.norm()), 1e-4 * gmax)))
    print(f"[{name}] worst per-parameter grad rel-L2 diff oracle vs reference {worst:.2e} over {len(g_ref)} params "
          f"({time.time() - t0:.1f}s)")
    assert worst < 1e-3, worst
    unused = sorted(k for k, p in named.items() if p.requires_grad and p.grad is None)
    small = {}
    for k, g in g_ref.items():
        if g.numel() <= 4096:
            small["grad:" + k] = g.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), logits=y_ref.detach().numpy().astype(np.float32),
                        loss=np.float64(loss_ref.item()),
                        meta=np.array(json.dumps(dict(model=model_name, B=B, H=H, W=W, num_classes=num_classes, mode="train",
                                                      grad_norms=norms, unused=unused))), **small)


class pin_reference_randomness:
    """Feed fixed draws to the reference's own train-mode random ops.  RoPE: `torch.empty(1).uniform_(lo, hi)` of the rescale
    (LAY/rope_position_encoding.py:96) returns the next pinned log-scale (the ViT runs once, under no_grad).  DropPath: the extractors
    run under torch.utils.checkpoint (ADP:151), which RE-RUNS their forward during backward with fresh bernoulli_ draws unless the RNG
    stream is the only source of randomness -- so each DropPath instance (ADP:29-37) gets its pinned mask as a per-module constant:
    forward(x) = x * mask, mask = Bernoulli(keep) / keep drawn once (exactly what drop_path :18-26 computes for that draw)."""

    def __init__(self, net, log_scales, masks):
        self.ls, self.i = [float(v) for v in log_scales], 0
        self.dps = [m for m in net.modules() if m.__class__.__name__ == "DropPath"]
        assert len(self.dps) == len(masks), (len(self.dps), len(masks))
        self.masks = [m.clone() for m in masks]

    def __enter__(self):
        self._u = torch.Tensor.uniform_
        outer = self

        def uniform_(t, a=0.0, b=1.0, **kw):
            assert t.numel() == 1, t.shape
            v = outer.ls[outer.i]; outer.i += 1
            assert a - 1e-6 <= v <= b + 1e-6
            return t.fill_(v)

        torch.Tensor.uniform_ = uniform_
        for m, mk in zip(self.dps, self.masks):
            m.forward = (lambda x, mk=mk: x * mk.to(x.dtype).view((-1,) + (1,) * (x.ndim - 1)))
        return self

    def __exit__(self, *a):
        torch.Tensor.uniform_ = self._u
        for m in self.dps:
            del m.forward


def train_case_pinned(name, model_name, B, H, W, num_classes=2, sample=65536, check_oracle=True):
    """train() forward + DC/CE + backward of the REFERENCE with its random ops pinned (per-block RoPE rescale, DropPath masks):
    golden = logits, loss, per-parameter gradient norms, and every gradient in full (<= `sample` elements) or as a fixed
    `sample`-element subset (weights.sample_indices)."""
    t0 = time.time()
    net, ks, sd = build(model_name, num_classes)
    net.train()
    depth = len(net.encoder.dinov3_adapter.backbone.blocks)
    log_scales, masks = weights.pinned_randomness(depth, B, seed=2)
    x = weights.make_input(B, 3, H, W, seed=2)
    tgt = weights.make_target(B, H, W, num_classes, seed=2)
    with pin_reference_randomness(net, log_scales, masks) as pr:
        y_ref = net(x)
        assert pr.i == depth, pr.i
        loss_ref = O.dc_and_ce_loss(y_ref, tgt)
        loss_ref.backward()                       # inside the context: checkpointed extractors re-run their forward here
    named = dict(net.named_parameters())
    g_ref = {k: p.grad for k, p in named.items() if p.requires_grad and p.grad is not None}
    print(f"[{name}] reference train step done ({time.time() - t0:.1f}s), loss {loss_ref.item():.6f}")
    if check_oracle:
        with torch.no_grad():
            y_or = O.dinounet_forward(x, sd, model_name, training=True, rope_rescale=log_scales.exp(), drop_masks=masks)
        e = rel(y_or, y_ref.detach())
        print(f"[{name}] oracle (pinned randomness) vs reference logits rel err {e:.2e}")
        assert e < TOL, e
    norms = {k: float(g.norm()) for k, g in g_ref.items()}
    unused = sorted(k for k, p in named.items() if p.requires_grad and p.grad is None)
    out = {}
    tot = 0
    for k, g in g_ref.items():
        flat = g.flatten()
        if flat.numel() > sample:
            flat = flat[weights.sample_indices(k, flat.numel(), sample)]
        out["grad:" + k] = flat.numpy().astype(np.float32)
        tot += flat.numel()
    print(f"[{name}] {len(g_ref)} gradients, {tot / 1e6:.2f} M elements stored")
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), logits=y_ref.detach().numpy().astype(np.float32),
                        loss=np.float64(loss_ref.item()),
                        meta=np.array(json.dumps(dict(model=model_name, B=B, H=H, W=W, num_classes=num_classes, mode="train_pinned",
                                                      sample=sample, grad_norms=norms, unused=unused))), **out)


VIT7B_STYLE = dict(embed_dim=256, depth=3, num_heads=2, ffn_ratio=3.0, qkv_bias=False, drop_path_rate=0.4, ffn_layer="swiglu64",
                   n_storage_tokens=4, mask_k_bias=True, untie_global_and_local_cls_norm=True)


def vit7b_style_case(name="vit7b_style_64"):
    """The code paths only the 7B backbone takes (hub/backbones.py:452-496), on the reference's own DinoVisionTransformer with 7B-style
    hyper-parameters at toy width: head dim 128, no qkv bias under a K-masked-bias Linear, SwiGLU-64 FFN (ffn_layers.py:52-77), and the
    train-mode batch-subset stochastic depth (layers/block.py:126-187) with rate 0.4.  get_intermediate_layers (the adapter's call,
    ADP:424-426) in eval mode and in train mode with the random ops pinned (per-block RoPE rescale draws, torch.randperm subsets)."""
    refshim.install()
    from dinounet.dinov3.models.vision_transformer import DinoVisionTransformer as RefViT
    cfg = VIT7B_STYLE
    torch.manual_seed(0)
    net = RefViT(patch_size=16, pos_embed_rope_base=100, pos_embed_rope_normalize_coords="separate", pos_embed_rope_rescale_coords=2,
                 pos_embed_rope_dtype="fp32", layerscale_init=1.0e-5, norm_layer="layernormbf16", ffn_bias=True, proj_bias=True, **cfg)
    net.init_weights()
    ks = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    sd = weights.make_state_dict(ks, seed=0)
    net.load_state_dict(sd, strict=True)
    B, depth = 5, cfg["depth"]
    x = weights.make_input(B, 3, 64, 64, seed=6)
    idx = list(range(depth))
    ocfg = dict(embed_dim=cfg["embed_dim"], depth=depth, num_heads=cfg["num_heads"], ffn="swiglu", qkv_bias=False, interaction_indexes=idx)
    out = {"meta": np.array(json.dumps(dict(cfg=cfg, B=B, H=64, W=64, keys=[[k, list(s_)] for k, s_ in ks])))}
    net.eval()
    with torch.no_grad():
        ref = net.get_intermediate_layers(x, n=idx, return_class_token=True)
        orc = O.vit_intermediate(x, O.SD(sd), ocfg)
    for i, ((rp, rc), (op, oc)) in enumerate(zip(ref, orc)):
        assert rel(op, rp) < TOL and rel(oc, rc) < TOL, i
        out[f"eval_patch{i}"], out[f"eval_cls{i}"] = rp.numpy().astype(np.float32), rc.numpy().astype(np.float32)
    net.train()
    log_scales, _ = weights.pinned_randomness(depth, B, seed=3)
    subsets = weights.pinned_subsets(depth, B, cfg["drop_path_rate"], seed=3)
    perms = [t for pair in subsets for t in pair]
    real_randperm = torch.randperm
    state = {"i": 0}

    def randperm(n, **kw):            # the k-prefix of the pinned permutation is what block.py:97,113 slices off
        t = perms[state["i"]]; state["i"] += 1
        rest = torch.tensor([j for j in range(n) if j not in t.tolist()], dtype=torch.int64)
        return torch.cat([t, rest])

    class _Pin(pin_reference_randomness):
        pass

    torch.randperm = randperm
    try:
        with torch.no_grad(), pin_reference_randomness(net, log_scales, []) as pr:
            ref = net.get_intermediate_layers(x, n=idx, return_class_token=True)
            assert pr.i == depth and state["i"] == 2 * depth, (pr.i, state["i"])
    finally:
        torch.randperm = real_randperm
    with torch.no_grad():
        orc = O.vit_intermediate(x, O.SD(sd), ocfg, rope_rescale=log_scales.exp(), subsets=subsets)
    for i, ((rp, rc), (op, oc)) in enumerate(zip(ref, orc)):
        assert rel(op, rp) < TOL and rel(oc, rc) < TOL, (i, rel(op, rp))
        out[f"train_patch{i}"], out[f"train_cls{i}"] = rp.numpy().astype(np.float32), rc.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print(f"[{name}] 7B-style ViT (Dh 128, SwiGLU-64, subset stochastic depth 0.4): oracle == reference in eval and pinned train mode (<{TOL})")


def msda_cases():
    """ops/test.py fixture (N,M,D=1,2,2; Lq,L,P=2,2,2; shapes (6,4),(3,2); manual_seed(3)) evaluated with the
    reference's own ms_deform_attn_core_pytorch in fp64 + its gradients, for the gradcheck channel list."""
    refshim.install()
    from dinounet.dinov3.eval.segmentation.models.utils.ms_deform_attn import ms_deform_attn_core_pytorch as core
    N, M, Lq, L, P = 1, 2, 2, 2, 2
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    torch.manual_seed(3)
    out = {"shapes": shapes.numpy(), "level_start_index": lsi.numpy()}
    for D in [2, 4, 12, 24, 30, 32, 64, 71, 128]:
        value = (torch.rand(N, S, M, D) * 0.01).double().requires_grad_(True)
        loc = torch.rand(N, Lq, M, L, P, 2).double().requires_grad_(True)
        aw = torch.rand(N, Lq, M, L, P) + 1e-5
        aw = (aw / aw.sum(-1, keepdim=True).sum(-2, keepdim=True)).double().requires_grad_(True)
        o = core(value, shapes, loc, aw)
        go = torch.rand(o.shape, dtype=torch.float64)
        gv, gl, ga = torch.autograd.grad(o, (value, loc, aw), go)
        o2 = O.msda_core_loops(value, shapes, lsi, loc, aw)
        assert (o2 - o.detach()).abs().max() < 1e-12, "scalar-loop restatement disagrees with reference core"
        o3 = O.msda_core(value.detach(), shapes.tolist(), loc.detach(), aw.detach())
        assert (o3 - o.detach()).abs().max() < 1e-12
        for n_, t in (("value", value), ("loc", loc), ("attn", aw), ("out", o), ("grad_out", go), ("grad_value", gv),
                      ("grad_loc", gl), ("grad_attn", ga)):
            out[f"D{D}_{n_}"] = t.detach().numpy()
    # the wide channel counts of the reference's own gradcheck (ops/test.py:120: 1025, 2048, 3096), same fixture, stored as fp32
    torch.manual_seed(4)
    for D in [1025, 2048, 3096]:
        value = (torch.rand(N, S, M, D) * 0.01).double().requires_grad_(True)
        loc = torch.rand(N, Lq, M, L, P, 2).double().requires_grad_(True)
        aw = torch.rand(N, Lq, M, L, P) + 1e-5
        aw = (aw / aw.sum(-1, keepdim=True).sum(-2, keepdim=True)).double().requires_grad_(True)
        o = core(value, shapes, loc, aw)
        go = torch.rand(o.shape, dtype=torch.float64)
        gv, gl, ga = torch.autograd.grad(o, (value, loc, aw), go)
        o3 = O.msda_core(value.detach(), shapes.tolist(), loc.detach(), aw.detach())
        assert (o3 - o.detach()).abs().max() < 1e-12
        for n_, t in (("value", value), ("loc", loc), ("attn", aw), ("out", o), ("grad_out", go), ("grad_value", gv),
                      ("grad_loc", gl), ("grad_attn", ga)):
            out[f"D{D}_{n_}"] = t.detach().numpy().astype(np.float32)
    # out-of-range / border sampling: locations in [-0.3, 1.3] exercise every zero-padding branch (cuh:293, :60-84)
    torch.manual_seed(5)
    D = 8
    value = torch.randn(2, S, M, D, dtype=torch.float64, requires_grad=True)
    loc = (torch.rand(2, 5, M, L, P, 2, dtype=torch.float64) * 1.6 - 0.3).requires_grad_(True)
    aw = torch.rand(2, 5, M, L, P, dtype=torch.float64).requires_grad_(True)
    o = core(value, shapes, loc, aw)
    go = torch.randn(o.shape, dtype=torch.float64)
    gv, gl, ga = torch.autograd.grad(o, (value, loc, aw), go)
    o2 = O.msda_core_loops(value, shapes, lsi, loc, aw)
    assert (o2 - o.detach()).abs().max() < 1e-12
    for n_, t in (("value", value), ("loc", loc), ("attn", aw), ("out", o), ("grad_out", go), ("grad_value", gv),
                  ("grad_loc", gl), ("grad_attn", ga)):
        out[f"border_{n_}"] = t.detach().numpy()
    np.savez_compressed(os.path.join(GOLD, "msda_testpy.npz"), **out)
    print("[msda] ops/test.py fixture + border case written; loop restatement == reference core (fp64, <1e-12)")


def main():
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["msda", "s", "b", "l", "train", "s512", "misc", "b512", "l512", "pinned", "vit7b"]
    if "msda" in which:
        msda_cases()
    if "s" in which:
        ks, _ = eval_case("dinounet_s_64_eval", "dinounet_s", 2, 3, 64, 64, with_taps=True)
        net, ks, _ = build("dinounet_s"); dump_keys("dinounet_s", ks, net.state_dict())
    if "misc" in which:
        eval_case("dinounet_s_96x64_eval", "dinounet_s", 1, 3, 96, 64)
        eval_case("dinounet_s_64_c1_eval", "dinounet_s", 1, 1, 64, 64)
        eval_case("dinounet_s_64_k4_eval", "dinounet_s", 1, 3, 64, 64, num_classes=4)
    if "train" in which:
        train_case("dinounet_s_64_train", "dinounet_s", 2, 64, 64)
    if "b" in which:
        eval_case("dinounet_b_64_eval", "dinounet_b", 1, 3, 64, 64)
        net, ks, _ = build("dinounet_b"); dump_keys("dinounet_b", ks, net.state_dict())
    if "l" in which:
        eval_case("dinounet_l_64_eval", "dinounet_l", 1, 3, 64, 64)
        net, ks, _ = build("dinounet_l"); dump_keys("dinounet_l", ks, net.state_dict())
    if "s512" in which:
        eval_case("dinounet_s_512_eval", "dinounet_s", 1, 3, 512, 512)
    if "b512" in which:     # BASELINE.json configs 2-3 at their own resolution (batch 1)
        eval_case("dinounet_b_512_eval", "dinounet_b", 1, 3, 512, 512)
    if "l512" in which:     # the headline config's shape: N = 1029 tokens, Lq = 5376 queries, 512^2 decoder
        eval_case("dinounet_l_512_eval", "dinounet_l", 1, 3, 512, 512)
    if "vit7b" in which:
        vit7b_style_case()
    if "pinned" in which:   # train mode with the random ops pinned on both sides
        train_case_pinned("dinounet_s_64_train_pinned", "dinounet_s", 2, 64, 64, sample=4096)
        train_case_pinned("dinounet_l_256_train_pinned", "dinounet_l", 2, 256, 256, sample=16384)
    if "pinned512" in which:   # the same at a BASELINE.json shape (512 x 512: N = 1029 tokens, Lq = 5376, 512^2 decoder), round 3
        train_case_pinned("dinounet_s_512_train_pinned", "dinounet_s", 2, 512, 512, sample=8192)
    if "pinned512l" in which:  # the HEADLINE model at its own resolution (dinounet_l, 512 x 512, batch 1), round 4
        train_case_pinned("dinounet_l_512_train_pinned", "dinounet_l", 1, 512, 512, sample=8192)


if __name__ == "__main__":
    main()
