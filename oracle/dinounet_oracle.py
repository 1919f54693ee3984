"""TEST INFRASTRUCTURE ONLY -- CPU restatement (plain PyTorch fp32, functional) of the reference's
Dino U-Net forward hot path.  It is the checker for the HIP path; it is never the thing shipped or
measured (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it).

Every function cites the reference file:line it restates (paths relative to /root/reference):
  DT  = dinounet_training.py
  ADP = dinounet/dinov3/eval/segmentation/models/backbone/dinov3_adapter.py
  MSA = dinounet/dinov3/eval/segmentation/models/utils/ms_deform_attn.py
  VIT = dinounet/dinov3/models/vision_transformer.py
  LAY = dinounet/dinov3/layers/
  HUB = dinounet/dinov3/hub/backbones.py

Pinned: oracle/make_golden.py imports the reference's own modules (through oracle/refshim.py) in the
build container, loads the same seeded state_dict into both, and asserts this restatement reproduces
the reference's logits and per-stage tensors (fp32, <=2e-5 rel); the reference outputs are committed
under tests/golden/.  The conv-block composition of the un-vendored `dynamic_network_architectures`
package is "parity unpinned" (SURVEY.md 8c): its source is not in the reference tree.

All arithmetic is differentiable torch, so `torch.autograd` through this file is the gradient oracle
(the reference's MSDA backward is gradchecked against exactly this formula, ops/test.py:101-111).
"""
import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

# ---- model table: HUB:201-236 (s), HUB:279-316 (b), HUB:318-360 (l), HUB:452-496 (7b); DT:36-48 ----
MODELS = {
    "dinounet_s": dict(embed_dim=384, depth=12, num_heads=6, ffn="mlp", ffn_hidden=1536, qkv_bias=True,
                       interaction_indexes=[2, 5, 8, 11]),
    "dinounet_b": dict(embed_dim=768, depth=12, num_heads=12, ffn="mlp", ffn_hidden=3072, qkv_bias=True,
                       interaction_indexes=[2, 5, 8, 11]),
    "dinounet_l": dict(embed_dim=1024, depth=24, num_heads=16, ffn="mlp", ffn_hidden=4096, qkv_bias=True,
                       interaction_indexes=[4, 11, 17, 23]),
    "dinounet_7b": dict(embed_dim=4096, depth=40, num_heads=32, ffn="swiglu", ffn_hidden=8192, qkv_bias=False,
                        interaction_indexes=[9, 19, 29, 39]),
}
N_STORAGE = 4          # HUB:230
PATCH = 16
ROPE_BASE = 100.0      # HUB:214
VIT_LN_EPS = 1e-5      # VIT:29 "layernormbf16"
ADP_LN_EPS = 1e-6      # ADP:348
DEFORM_HEADS = 16      # DT:760
DEFORM_POINTS = 4      # DT:759
FAPM_RANK = 256        # DT:449


class SD:
    """state_dict view with a key prefix."""

    def __init__(self, sd: Dict[str, torch.Tensor], prefix: str = ""):
        self.sd, self.prefix = sd, prefix

    def __getitem__(self, k):
        return self.sd[self.prefix + k]

    def has(self, k):
        return (self.prefix + k) in self.sd

    def sub(self, p):
        return SD(self.sd, self.prefix + p)


# ----------------------------------------------------------------------------------------------
# ViT backbone (frozen)
# ----------------------------------------------------------------------------------------------
def rope_sincos(H: int, W: int, periods: torch.Tensor, rescale: Optional[float] = None):
    """LAY/rope_position_encoding.py:57-106 (normalize_coords="separate", fp32).  `rescale` is the
    train-mode log-uniform factor of :93-97 (host-drawn scalar); None in eval."""
    dd = dict(dtype=torch.float32)
    coords_h = torch.arange(0.5, H, **dd) / H
    coords_w = torch.arange(0.5, W, **dd) / W
    coords = torch.stack(torch.meshgrid(coords_h, coords_w, indexing="ij"), dim=-1).flatten(0, 1)
    coords = 2.0 * coords - 1.0
    if rescale is not None:
        coords = coords * rescale
    angles = 2 * math.pi * coords[:, :, None] / periods[None, None, :].float()
    angles = angles.flatten(1, 2).tile(2)
    return torch.sin(angles), torch.cos(angles)


def rope_periods(d_head: int):
    """LAY/rope_position_encoding.py:108-121 (base parametrisation)."""
    return ROPE_BASE ** (2 * torch.arange(d_head // 4, dtype=torch.float32) / (d_head // 2))


def _rotate_half(x):  # LAY/attention.py:16-20
    x1, x2 = x.chunk(2, dim=-1)
    return torch.cat([-x2, x1], dim=-1)


def vit_attention(x, p: SD, num_heads: int, sin, cos):
    """LAY/attention.py:87-118 (+ LinearKMaskedBias :30-40, apply_rope :66-85)."""
    B, N, C = x.shape
    bias = None
    if p.has("qkv.bias"):
        bias = p["qkv.bias"]
        if p.has("qkv.bias_mask"):
            bias = bias * p["qkv.bias_mask"].to(bias.dtype)
    qkv = F.linear(x, p["qkv.weight"], bias).reshape(B, N, 3, num_heads, C // num_heads)
    q, k, v = [t.transpose(1, 2) for t in torch.unbind(qkv, 2)]
    prefix = N - sin.shape[-2]
    q = torch.cat((q[:, :, :prefix], q[:, :, prefix:] * cos + _rotate_half(q[:, :, prefix:]) * sin), dim=-2)
    k = torch.cat((k[:, :, :prefix], k[:, :, prefix:] * cos + _rotate_half(k[:, :, prefix:]) * sin), dim=-2)
    o = F.scaled_dot_product_attention(q, k, v)
    o = o.transpose(1, 2).reshape(B, N, C)
    return F.linear(o, p["proj.weight"], p["proj.bias"])


def vit_ffn(x, p: SD, kind: str):
    """LAY/ffn_layers.py:43-49 (Mlp, erf-GELU) / :73-77 (SwiGLU)."""
    if kind == "mlp":
        return F.linear(F.gelu(F.linear(x, p["fc1.weight"], p["fc1.bias"])), p["fc2.weight"], p["fc2.bias"])
    h = F.silu(F.linear(x, p["w1.weight"], p["w1.bias"])) * F.linear(x, p["w2.weight"], p["w2.bias"])
    return F.linear(h, p["w3.weight"], p["w3.bias"])


def vit_block(x, p: SD, cfg, sin, cos, subset=None):
    """LAY/block.py:189-194 (eval / drop-path 0 branch); LayerScale LAY/layer_scale.py:28.
    subset = (idx_attn, idx_ffn): the train-mode batch-subset stochastic depth of LAY/block.py:126-187 -- each branch runs on the
    samples idx only and its LayerScale-d output is added back scaled by B / len(idx) (torch.index_add with alpha)."""
    D = x.shape[-1]

    def attn(t):
        h = F.layer_norm(t, (D,), p["norm1.weight"], p["norm1.bias"], VIT_LN_EPS)
        return p["ls1.gamma"] * vit_attention(h, p.sub("attn."), cfg["num_heads"], sin, cos)

    def ffn(t):
        h = F.layer_norm(t, (D,), p["norm2.weight"], p["norm2.bias"], VIT_LN_EPS)
        return p["ls2.gamma"] * vit_ffn(h, p.sub("mlp."), cfg["ffn"])

    if subset is None:
        x = x + attn(x)
        return x + ffn(x)
    i1, i2 = subset
    scale = x.shape[0] / i1.numel()
    x = torch.index_add(x, 0, i1, attn(x[i1]), alpha=scale)
    return torch.index_add(x, 0, i2, ffn(x[i2]), alpha=scale)


def vit_intermediate(x, p: SD, cfg, rope_rescale=None, subsets=None):
    """VIT:281-318 get_intermediate_layers(n=interaction_indexes, return_class_token=True, norm=True)
    via VIT:265-279 and prepare_tokens_with_masks VIT:186-216.  Returns [(patch (B,hw,D), cls (B,D))]."""
    B = x.shape[0]
    D = cfg["embed_dim"]
    t = F.conv2d(x, p["patch_embed.proj.weight"], p["patch_embed.proj.bias"], stride=PATCH)  # LAY/patch_embed.py:70
    H, W = t.shape[2], t.shape[3]
    t = t.flatten(2).transpose(1, 2)
    cls = p["cls_token"] + 0 * p["mask_token"]                                              # VIT:195
    t = torch.cat([cls.expand(B, -1, -1), p["storage_tokens"].expand(B, -1, -1), t], dim=1)
    per_block = isinstance(rope_rescale, (list, tuple)) or (torch.is_tensor(rope_rescale) and rope_rescale.dim() == 1)
    if not per_block:
        sin, cos = rope_sincos(H, W, p["rope_embed.periods"], rope_rescale)
    outs = []
    for i in range(cfg["depth"]):
        if per_block:                                                                        # train mode: one draw per block, VIT:271-272
            sin, cos = rope_sincos(H, W, p["rope_embed.periods"], float(rope_rescale[i]))
        t = vit_block(t, p.sub(f"blocks.{i}."), cfg, sin, cos, None if subsets is None else subsets[i])
        if i in cfg["interaction_indexes"]:
            o = F.layer_norm(t, (D,), p["norm.weight"], p["norm.bias"], VIT_LN_EPS)          # VIT:300
            outs.append((o[:, N_STORAGE + 1:], o[:, 0]))
    return outs


# ----------------------------------------------------------------------------------------------
# MSDeformAttn
# ----------------------------------------------------------------------------------------------
def msda_core(value, spatial_shapes, sampling_locations, attention_weights):
    """MSA:71-92 ms_deform_attn_core_pytorch (grid_sample formulation; == the CUDA forward kernel,
    ops/src/cuda/ms_deform_im2col_cuda.cuh:242-304).
    value (N,S,M,D); sampling_locations (N,Lq,M,L,P,2) as (x,y) in [0,1]; weights (N,Lq,M,L,P)."""
    N_, S_, M_, D_ = value.shape
    _, Lq_, _, L_, P_, _ = sampling_locations.shape
    shapes = [(int(h), int(w)) for h, w in spatial_shapes]
    value_list = value.split([h * w for h, w in shapes], dim=1)
    grids = 2 * sampling_locations - 1
    sampled = []
    for lid, (H_, W_) in enumerate(shapes):
        v = value_list[lid].flatten(2).transpose(1, 2).reshape(N_ * M_, D_, H_, W_)
        g = grids[:, :, :, lid].transpose(1, 2).flatten(0, 1)
        sampled.append(F.grid_sample(v, g, mode="bilinear", padding_mode="zeros", align_corners=False))
    aw = attention_weights.transpose(1, 2).reshape(N_ * M_, 1, Lq_, L_ * P_)
    out = (torch.stack(sampled, dim=-2).flatten(-2) * aw).sum(-1).view(N_, M_ * D_, Lq_)
    return out.transpose(1, 2).contiguous()


def msda_core_loops(value, spatial_shapes, level_start_index, loc, attn):
    """Scalar restatement of the CUDA forward loop (cuh:258-303 + bilinear cuh:38-89) in numpy float64:
    pixel = loc*size - 0.5, in-range test (-1, size), per-corner zero padding.  Small cases only."""
    import numpy as np
    v = value.detach().double().numpy(); lo = loc.detach().double().numpy(); aw = attn.detach().double().numpy()
    N, S, M, D = v.shape
    _, Lq, _, L, P, _ = lo.shape
    out = np.zeros((N, Lq, M, D))
    for b in range(N):
        for q in range(Lq):
            for m in range(M):
                for l in range(L):
                    Hh, Ww = int(spatial_shapes[l][0]), int(spatial_shapes[l][1])
                    start = int(level_start_index[l])
                    for p in range(P):
                        w_im = lo[b, q, m, l, p, 0] * Ww - 0.5
                        h_im = lo[b, q, m, l, p, 1] * Hh - 0.5
                        if not (h_im > -1 and w_im > -1 and h_im < Hh and w_im < Ww):
                            continue
                        h0, w0 = math.floor(h_im), math.floor(w_im)
                        lh, lw = h_im - h0, w_im - w0
                        acc = np.zeros(D)
                        for (hh, ww, wt) in ((h0, w0, (1 - lh) * (1 - lw)), (h0, w0 + 1, (1 - lh) * lw),
                                             (h0 + 1, w0, lh * (1 - lw)), (h0 + 1, w0 + 1, lh * lw)):
                            if 0 <= hh <= Hh - 1 and 0 <= ww <= Ww - 1:
                                acc += wt * v[b, start + hh * Ww + ww, m]
                        out[b, q, m] += aw[b, q, m, l, p] * acc
    return torch.from_numpy(out.reshape(N, Lq, M * D))


def msda_backward(value, spatial_shapes, loc, attn, grad_output):
    """Gradient oracle for MSA:50-68 / cu:88-158: autograd through msda_core (ops/test.py:101-111)."""
    with torch.enable_grad():
        v = value.detach().requires_grad_(True)
        l = loc.detach().requires_grad_(True)
        a = attn.detach().requires_grad_(True)
        out = msda_core(v, spatial_shapes, l, a)
        return torch.autograd.grad(out, (v, l, a), grad_output)


def msdeform_attn(query, reference_points, feat, spatial_shapes, p: SD):
    """MSA:158-216 (reference_points last dim 2; n_levels=1, 16 heads, 4 points, ratio 0.5)."""
    N, Lq, C = query.shape
    _, Lin, _ = feat.shape
    M, L, P = DEFORM_HEADS, len(spatial_shapes), DEFORM_POINTS
    value = F.linear(feat, p["value_proj.weight"], p["value_proj.bias"])
    value = value.view(N, Lin, M, value.shape[-1] // M)
    off = F.linear(query, p["sampling_offsets.weight"], p["sampling_offsets.bias"]).view(N, Lq, M, L, P, 2)
    aw = F.linear(query, p["attention_weights.weight"], p["attention_weights.bias"]).view(N, Lq, M, L * P)
    aw = F.softmax(aw, -1).view(N, Lq, M, L, P)
    normalizer = torch.tensor([[float(w), float(h)] for h, w in spatial_shapes])           # MSA:193 (W,H)
    loc = reference_points[:, :, None, :, None, :] + off / normalizer[None, None, None, :, None, :]
    out = msda_core(value.float(), spatial_shapes, loc.float(), aw.float())
    return F.linear(out, p["output_proj.weight"], p["output_proj.bias"])


# ----------------------------------------------------------------------------------------------
# Adapter
# ----------------------------------------------------------------------------------------------
def reference_points(shapes):
    """ADP:40-53 get_reference_points -> (1, sum(HW), 1, 2) as (x, y) in (0,1)."""
    refs = []
    for (H_, W_) in shapes:
        ry, rx = torch.meshgrid(torch.linspace(0.5, H_ - 0.5, H_), torch.linspace(0.5, W_ - 0.5, W_), indexing="ij")
        refs.append(torch.stack((rx.reshape(-1)[None] / W_, ry.reshape(-1)[None] / H_), -1))
    return torch.cat(refs, 1)[:, :, None]


def _bn(x, p: SD, training: bool, eps=1e-5):
    """nn.SyncBatchNorm (ADP:242,361): batch statistics in train mode (running-stat update not
    modelled here), running statistics in eval."""
    if training:
        return F.batch_norm(x, None, None, p["weight"], p["bias"], True, 0.0, eps)
    return F.batch_norm(x, p["running_mean"], p["running_var"], p["weight"], p["bias"], False, 0.0, eps)


def spm(x, p: SD, training: bool):
    """ADP:279-302 SpatialPriorModule._inner_forward; ctor ADP:235-277."""
    c = F.relu(_bn(F.conv2d(x, p["stem.0.weight"], None, 2, 1), p.sub("stem.1."), training))
    c = F.relu(_bn(F.conv2d(c, p["stem.3.weight"], None, 1, 1), p.sub("stem.4."), training))
    c = F.relu(_bn(F.conv2d(c, p["stem.6.weight"], None, 1, 1), p.sub("stem.7."), training))
    c1 = F.max_pool2d(c, 3, 2, 1)
    c2 = F.relu(_bn(F.conv2d(c1, p["conv2.0.weight"], None, 2, 1), p.sub("conv2.1."), training))
    c3 = F.relu(_bn(F.conv2d(c2, p["conv3.0.weight"], None, 2, 1), p.sub("conv3.1."), training))
    c4 = F.relu(_bn(F.conv2d(c3, p["conv4.0.weight"], None, 2, 1), p.sub("conv4.1."), training))
    c1 = F.conv2d(c1, p["fc1.weight"], p["fc1.bias"])
    c2 = F.conv2d(c2, p["fc2.weight"], p["fc2.bias"])
    c3 = F.conv2d(c3, p["fc3.weight"], p["fc3.bias"])
    c4 = F.conv2d(c4, p["fc4.weight"], p["fc4.bias"])
    bs, dim = c1.shape[:2]
    tok = lambda t: t.view(bs, dim, -1).transpose(1, 2)
    return c1, tok(c2), tok(c3), tok(c4)


def conv_ffn(x, H, W, p: SD):
    """ADP:84-91 ConvFFN + ADP:99-109 DWConv (one depthwise 3x3 shared by the three token grids)."""
    x = F.linear(x, p["fc1.weight"], p["fc1.bias"])
    B, N, C = x.shape
    n = N // 21
    w, b = p["dwconv.dwconv.weight"], p["dwconv.dwconv.bias"]
    outs = []
    for (lo, hi, h, ww) in ((0, 16 * n, H * 2, W * 2), (16 * n, 20 * n, H, W), (20 * n, N, H // 2, W // 2)):
        t = x[:, lo:hi].transpose(1, 2).reshape(B, C, h, ww)
        outs.append(F.conv2d(t, w, b, 1, 1, groups=C).flatten(2).transpose(1, 2))
    x = F.gelu(torch.cat(outs, dim=1))
    return F.linear(x, p["fc2.weight"], p["fc2.bias"])


def extractor(query, ref, feat, shapes, H, W, p: SD, drop_mask=None):
    """ADP:140-156 Extractor._inner_forward (activation checkpointing ADP:151 does not change values).
    drop_mask: per-sample keep/keep_prob factors (B,) for DropPath ADP:18-26 in train mode, else None."""
    D = query.shape[-1]
    qn = F.layer_norm(query, (D,), p["query_norm.weight"], p["query_norm.bias"], ADP_LN_EPS)
    fn = F.layer_norm(feat, (D,), p["feat_norm.weight"], p["feat_norm.bias"], ADP_LN_EPS)
    query = query + msdeform_attn(qn, ref, fn, shapes, p.sub("attn."))
    f = conv_ffn(F.layer_norm(query, (D,), p["ffn_norm.weight"], p["ffn_norm.bias"], ADP_LN_EPS), H, W, p.sub("ffn."))
    if drop_mask is not None:
        f = f * drop_mask.view(-1, 1, 1)
    return query + f


def adapter(x, p: SD, cfg, training=False, rope_rescale=None, drop_masks=None, taps=None):
    """ADP:408-484 DINOv3_Adapter.forward.  `drop_masks`: list of 6 (B,) tensors or None."""
    bs, _, h, w = x.shape
    D = cfg["embed_dim"]
    H_c, W_c = h // 16, w // 16
    H_t, W_t = h // PATCH, w // PATCH
    ref2 = reference_points([(h // 8, w // 8), (h // 16, w // 16), (h // 32, w // 32)])       # ADP:67
    shapes2 = [(H_t, W_t)]                                                                    # ADP:65
    c1, c2, c3, c4 = spm(x, p.sub("spm."), training)
    n2, n3 = c2.shape[1], c3.shape[1]
    le = p["level_embed"]
    c = torch.cat([c2 + le[0], c3 + le[1], c4 + le[2]], dim=1)                                # ADP:413-415
    layers = vit_intermediate(x, p.sub("backbone."), cfg, rope_rescale)                       # ADP:422-426 (no_grad)
    layers = [(a.detach(), b.detach()) for a, b in layers]
    if taps is not None:
        taps["vit"] = [a for a, _ in layers]
        taps["c0"] = c
    outs, k = [], 0
    for i in range(4):
        xi, _ = layers[i]
        pi = p.sub(f"interactions.{i}.")
        dm = (lambda j: None if drop_masks is None else drop_masks[j])
        c = extractor(c, ref2, xi, shapes2, H_c, W_c, pi.sub("extractor."), dm(k)); k += 1
        if i == 3:                                                                            # ADP:220-230
            for e in range(2):
                c = extractor(c, ref2, xi, shapes2, H_c, W_c, pi.sub(f"extra_extractors.{e}."), dm(k)); k += 1
        outs.append(xi.transpose(1, 2).reshape(bs, D, H_t, W_t))
        if taps is not None:
            taps[f"c{i + 1}"] = c
    c2 = c[:, :n2].transpose(1, 2).reshape(bs, D, H_c * 2, W_c * 2)
    c3 = c[:, n2:n2 + n3].transpose(1, 2).reshape(bs, D, H_c, W_c)
    c4 = c[:, n2 + n3:].transpose(1, 2).reshape(bs, D, H_c // 2, W_c // 2)
    c1 = F.conv_transpose2d(c2, p["up.weight"], p["up.bias"], stride=2) + c1                  # ADP:467
    sizes = [(4 * H_c, 4 * W_c), (2 * H_c, 2 * W_c), (H_c, W_c), (H_c // 2, W_c // 2)]
    cs = [ci + F.interpolate(o, size=s, mode="bilinear", align_corners=False)                 # ADP:472-476
          for ci, o, s in zip((c1, c2, c3, c4), outs, sizes)]
    return [_bn(ci, p.sub(f"norm{j + 1}."), training) for j, ci in enumerate(cs)]            # ADP:479-482


# ----------------------------------------------------------------------------------------------
# FAPM / ups / decoder  (DT)
# ----------------------------------------------------------------------------------------------
def _in_lrelu(x, w, b, eps=1e-5, slope=0.01):
    return F.leaky_relu(F.instance_norm(x, None, None, w, b, True, 0.0, eps), slope)


def fapm(x_list, p: SD):
    """DT:419-441 FAPM.forward (norm=InstanceNorm2d eps 1e-5 affine, act=LeakyReLU 0.01 from the plans;
    DepthwiseSeparableConv DT:241-246, SqueezeExcitation DT:222-225)."""
    out = []
    for i, x in enumerate(x_list):
        zs = F.conv2d(x, p["shared_basis.weight"], p["shared_basis.bias"])
        zi = F.conv2d(x, p[f"specific_bases.{i}.weight"], p[f"specific_bases.{i}.bias"])
        gb = F.conv2d(zs, p[f"film_generators.{i}.weight"], p[f"film_generators.{i}.bias"])
        gamma, beta = torch.chunk(gb, 2, dim=1)
        z = gamma * zi + beta
        r = p.sub(f"refinement_blocks.{i}.")
        t = F.conv2d(z, r["0.weight"], r["0.bias"])
        t = _in_lrelu(t, r["1.weight"], r["1.bias"])
        oc = t.shape[1]
        t = F.conv2d(t, r["3.depthwise.weight"], r["3.depthwise.bias"], 1, 1, groups=oc)
        t = F.conv2d(t, r["3.pointwise.weight"], r["3.pointwise.bias"])
        t = _in_lrelu(t, r["3.bn.weight"], r["3.bn.bias"])
        t = F.conv2d(t, r["4.weight"], r["4.bias"])
        s = t.mean((2, 3), keepdim=True)
        s = torch.sigmoid(F.conv2d(F.relu(F.conv2d(s, r["5.fc.0.weight"], r["5.fc.0.bias"])),
                                   r["5.fc.2.weight"], r["5.fc.2.bias"]))
        t = t * s
        sk = f"shortcut_projections.{i}.weight"
        short = F.conv2d(z, p[sk], p[f"shortcut_projections.{i}.bias"]) if p.has(sk) else z
        out.append(t + short)
    return out


def encoder(x, p: SD, cfg, training=False, rope_rescale=None, drop_masks=None, taps=None):
    """DT:489-511 DINOv3EncoderAdapter.forward; LearnableUpsampleBlock DT:255-264."""
    B, C, H, W = x.shape
    if C == 1:
        x = x.repeat(1, 3, 1, 1)
    elif C != 3:
        x = x.repeat(1, 3 // C + (1 if 3 % C != 0 else 0), 1, 1)[:, :3] if C < 3 else x[:, :3]
    feats = adapter(x, p.sub("dinov3_adapter."), cfg, training, rope_rescale, drop_masks, taps)
    ys = fapm(feats, p.sub("fapm."))
    skips = []
    for i, y in enumerate(ys):
        th, tw = H // (2 ** i), W // (2 ** i)
        while y.shape[2] * 2 <= th and y.shape[3] * 2 <= tw:
            y = F.conv_transpose2d(y, p[f"ups.{i}.up2.weight"], p[f"ups.{i}.up2.bias"], stride=2)
        if (y.shape[2], y.shape[3]) != (th, tw):
            y = F.interpolate(y, size=(th, tw), mode="bilinear", align_corners=False)
        skips.append(y)
    if taps is not None:
        taps["feats"], taps["fapm"], taps["skips"] = feats, ys, skips
    return skips


def decoder(skips, p: SD, deep_supervision=False):
    """DT:603-629 UNetDecoder.forward; stage = 2x [conv3x3(bias) -> InstanceNorm(eps1e-5,affine) ->
    LeakyReLU(0.01)] (dynamic_network_architectures StackedConvBlocks, called at DT:581-592)."""
    lres = skips[-1]
    segs = []
    n = len(skips) - 1
    for s in range(n):
        x = F.conv_transpose2d(lres, p[f"transpconvs.{s}.weight"], p[f"transpconvs.{s}.bias"], stride=2)
        x = torch.cat((x, skips[-(s + 2)]), 1)
        for m in range(2):
            q = p.sub(f"stages.{s}.convs.{m}.")
            x = _in_lrelu(F.conv2d(x, q["conv.weight"], q["conv.bias"], 1, 1), q["norm.weight"], q["norm.bias"])
        if deep_supervision or s == n - 1:
            segs.append(F.conv2d(x, p[f"seg_layers.{s}.weight"], p[f"seg_layers.{s}.bias"]))
        lres = x
    segs = segs[::-1]
    return segs if deep_supervision else segs[0]


def dinounet_forward(x, sd: Dict[str, torch.Tensor], model_name: str, training=False, rope_rescale=None,
                     drop_masks=None, deep_supervision=False, taps=None):
    """DT:786-804 DinoUNet.forward = decoder(encoder(x))."""
    cfg = MODELS[model_name]
    p = SD(sd)
    skips = encoder(x, p.sub("encoder."), cfg, training, rope_rescale, drop_masks, taps)
    return decoder(skips, p.sub("decoder."), deep_supervision)


# ----------------------------------------------------------------------------------------------
# loss used by the training-step harness (dinounet/training/loss/compound_losses.py:8-56,
# dice.py:58-119 MemoryEfficientSoftDiceLoss with batch_dice=True, do_bg=False, smooth 1e-5, ddp off)
# ----------------------------------------------------------------------------------------------
def dc_and_ce_loss(logits, target, smooth=1e-5):
    K = logits.shape[1]
    ce = F.cross_entropy(logits, target[:, 0].long())
    prob = F.softmax(logits, 1)
    onehot = F.one_hot(target[:, 0].long(), K).permute(0, 3, 1, 2).to(prob.dtype)
    axes = (2, 3)
    inter = (prob * onehot).sum(axes)[:, 1:].sum(0)
    sum_pred = prob.sum(axes)[:, 1:].sum(0)
    sum_gt = onehot.sum(axes)[:, 1:].sum(0)
    dc = (2 * inter + smooth) / torch.clip(sum_gt + sum_pred + smooth, 1e-8)
    return ce - dc.mean()
