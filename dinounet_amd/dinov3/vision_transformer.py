"""Frozen DINOv3 ViT feature extractor on the HIP kernels (forward only).

Mirrors the module tree / state_dict of the reference `DinoVisionTransformer`
(dinounet/dinov3/models/vision_transformer.py:55-176) so upstream DINOv3 checkpoints load with strict=True
(dinounet_training.py:66-67), and the call used by the adapter (`get_intermediate_layers(x, n=idx,
return_class_token=True)`, dinov3_adapter.py:424-426).  The torch sub-modules are parameter containers only;
the arithmetic runs in libdinounet_hip.so:

  patch embed  : du_patchify16 + du_gemm                      (layers/patch_embed.py:61-76)
  per block    : du_layernorm_fwd -> du_gemm(QKV, K-masked bias; bf16: RoPE + head split in its epilogue, else du_qkv_rope_split) -> du_attention_fwd
                 -> du_gemm(proj, LayerScale + residual epilogue) -> du_layernorm_fwd
                 -> du_gemm(fc1 + erf-GELU epilogue) -> du_gemm(fc2, LayerScale + residual epilogue)
                                                              (layers/block.py:189-194, attention.py:87-118)
The residual stream stays fp32 in HBM, GEMM inputs are bf16 (fp32 in parity mode), accumulation fp32.
"""
import math
import os

import numpy as np
import torch
from torch import nn

from .. import ops
from .._lib import ACT_GELU

# hub/backbones.py:201-236 (s), :279-316 (b), :318-360 (l), :452-496 (7b)
VIT_CONFIGS = {
    "dinounet_s": dict(embed_dim=384, depth=12, num_heads=6, ffn_layer="mlp", ffn_ratio=4.0, qkv_bias=True),
    "dinounet_b": dict(embed_dim=768, depth=12, num_heads=12, ffn_layer="mlp", ffn_ratio=4.0, qkv_bias=True),
    "dinounet_l": dict(embed_dim=1024, depth=24, num_heads=16, ffn_layer="mlp", ffn_ratio=4.0, qkv_bias=True),
    "dinounet_7b": dict(embed_dim=4096, depth=40, num_heads=32, ffn_layer="swiglu64", ffn_ratio=3.0, qkv_bias=False,
                        untie_global_and_local_cls_norm=True, drop_path_rate=0.4),
}


class LinearKMaskedBias(nn.Linear):
    """layers/attention.py:30-40: bias multiplied by a {1,0,1} mask that zeroes the K third."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        if self.bias is not None:
            self.register_buffer("bias_mask", torch.full_like(self.bias, fill_value=math.nan))


class SelfAttention(nn.Module):
    def __init__(self, dim, num_heads, qkv_bias, mask_k_bias):
        super().__init__()
        self.num_heads = num_heads
        self.qkv = (LinearKMaskedBias if mask_k_bias else nn.Linear)(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim, bias=True)


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class SwiGLUFFN(nn.Module):
    """layers/ffn_layers.py:52-77 with align_to=64 ("swiglu64", vision_transformer.py:23)."""

    def __init__(self, dim, hidden, align_to=64):
        super().__init__()
        d = int(hidden * 2 / 3)
        h = d + (-d % align_to)
        self.w1 = nn.Linear(dim, h)
        self.w2 = nn.Linear(dim, h)
        self.w3 = nn.Linear(h, dim)


class LayerScale(nn.Module):
    def __init__(self, dim, init_values=1e-5):
        super().__init__()
        self.gamma = nn.Parameter(torch.full((dim,), float(init_values)))


class SelfAttentionBlock(nn.Module):
    def __init__(self, dim, num_heads, ffn_layer, ffn_ratio, qkv_bias, mask_k_bias):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-5)            # "layernormbf16", vision_transformer.py:29
        self.attn = SelfAttention(dim, num_heads, qkv_bias, mask_k_bias)
        self.ls1 = LayerScale(dim)
        self.norm2 = nn.LayerNorm(dim, eps=1e-5)
        hidden = int(dim * ffn_ratio)
        self.mlp = Mlp(dim, hidden) if ffn_layer == "mlp" else SwiGLUFFN(dim, hidden, 64)
        self.ls2 = LayerScale(dim)


class PatchEmbed(nn.Module):
    def __init__(self, patch_size, in_chans, embed_dim):
        super().__init__()
        self.patch_size = (patch_size, patch_size)
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)


class RopePositionEmbedding(nn.Module):
    """layers/rope_position_encoding.py:16-121 (base 100, normalize_coords "separate", fp32, rescale_coords 2)."""

    def __init__(self, embed_dim, num_heads, base=100.0, rescale_coords=2.0):
        super().__init__()
        self.D_head = embed_dim // num_heads
        self.base = base
        self.rescale_coords = rescale_coords
        self.pinned_log_scales = None       # test hook: fixed log rescale factors instead of the device RNG draw
        n = self.D_head // 4
        self.register_buffer("periods", base ** (2 * torch.arange(n, dtype=torch.float32) / (self.D_head // 2)), persistent=True)
        self._cache = {}

    def sincos(self, H, W, device, training):
        """Returns (sin, cos) (HW, D_head) fp32 on `device`.  In train mode the reference draws ONE log-uniform
        rescale factor per call (:93-97) -- i.e. per block, since the table is rebuilt in every block
        (vision_transformer.py:271-272).  Everything is computed on the device (device RNG, no host sync, graph-safe);
        the eval table is cached."""
        key = (H, W, str(device))
        c = self._cache.get(key)
        if c is None:
            coords_h = torch.arange(0.5, H, dtype=torch.float32) / H
            coords_w = torch.arange(0.5, W, dtype=torch.float32) / W
            coords = torch.stack(torch.meshgrid(coords_h, coords_w, indexing="ij"), dim=-1).flatten(0, 1)
            coords = (2.0 * coords - 1.0).to(device)
            base = 2 * math.pi * coords[:, :, None] / self.periods.detach().float().to(device)[None, None, :]   # (HW, 2, D/4)
            ang = base.flatten(1, 2).tile(2)
            c = (base, torch.sin(ang), torch.cos(ang))
            self._cache = {key: c}
        base, sin, cos = c
        if training and self.rescale_coords is not None:
            mx = float(np.log(self.rescale_coords))
            scale = torch.empty(1, device=device, dtype=torch.float32).uniform_(-mx, mx).exp()
            ang = (base * scale).flatten(1, 2).tile(2)
            return torch.sin(ang), torch.cos(ang)
        return sin, cos

    def sincos_all(self, H, W, device, training, nblocks):
        """Tables for every block of one forward at once: (nblocks, HW, D_head) sin and cos.  Train mode draws the per-block
        log-uniform rescale factors (:93-97) as ONE vector, so the jitter costs 6 launches per forward instead of 6 per block."""
        sin, cos = self.sincos(H, W, device, False)
        if not (training and self.rescale_coords is not None):
            return sin[None].expand(nblocks, -1, -1), cos[None].expand(nblocks, -1, -1)
        base = self._cache[(H, W, str(device))][0]
        mx = float(np.log(self.rescale_coords))
        if self.pinned_log_scales is not None:      # parity tests feed the draws the reference was given (one per block)
            scale = self.pinned_log_scales.to(device=device, dtype=torch.float32).exp()
            assert scale.numel() == nblocks
        else:
            scale = torch.empty(nblocks, device=device, dtype=torch.float32).uniform_(-mx, mx).exp()
        ang = (base[None] * scale[:, None, None, None]).flatten(2, 3).tile(2)
        return torch.sin(ang), torch.cos(ang)


class DinoVisionTransformer(nn.Module):
    def __init__(self, *, patch_size=16, in_chans=3, embed_dim=768, depth=12, num_heads=12, ffn_ratio=4.0, qkv_bias=True,
                 ffn_layer="mlp", n_storage_tokens=4, mask_k_bias=True, untie_global_and_local_cls_norm=False,
                 rope_rescale_coords=2.0, drop_path_rate=0.0, **ignored):
        super().__init__()
        # train-mode batch-subset stochastic depth of every block (layers/block.py:126-187; 0.4 for the 7B model, hub/backbones.py:480,
        # 0 otherwise).  The backbone is frozen but nnUNetTrainer.py:890 puts the whole network in train(), so the reference applies it.
        self.drop_path_rate = float(drop_path_rate)
        self.pinned_subsets = None          # test hook: [(idx_attn, idx_ffn)] per block instead of torch.randperm draws
        self.num_features = self.embed_dim = embed_dim
        self.n_blocks = depth
        self.num_heads = num_heads
        self.patch_size = patch_size
        self.ffn_layer = ffn_layer
        self.patch_embed = PatchEmbed(patch_size, in_chans, embed_dim)
        self.cls_token = nn.Parameter(torch.empty(1, 1, embed_dim))
        self.n_storage_tokens = n_storage_tokens
        if n_storage_tokens > 0:
            self.storage_tokens = nn.Parameter(torch.empty(1, n_storage_tokens, embed_dim))
        self.rope_embed = RopePositionEmbedding(embed_dim, num_heads, rescale_coords=rope_rescale_coords)
        self.blocks = nn.ModuleList([SelfAttentionBlock(embed_dim, num_heads, ffn_layer, ffn_ratio, qkv_bias, mask_k_bias)
                                     for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-5)
        self.local_cls_norm = nn.LayerNorm(embed_dim, eps=1e-5) if untie_global_and_local_cls_norm else None
        self.head = nn.Identity()
        self.mask_token = nn.Parameter(torch.empty(1, embed_dim))
        self._cache = None
        self._ws = {}
        # Half-batch chains on side HIP streams (round 5).  The blocks of a frozen ViT are a strictly sequential chain of
        # launches, each with one workgroup per CU (128-147 KB of LDS): every launch boundary drains the chip -- cold prologue burst of 256
        # workgroups at once, a ragged last round (qkv: 1.5 rounds of 256 x 256 tiles), an exposed epilogue.  The samples of a batch are
        # independent, so the same work as TWO chains lets the dispatcher fill the CUs one chain leaves idle with the other chain's
        # workgroups; nothing else changes (same kernels, same arithmetic per sample).  Measured (profiles/r05_ab_*): the backbone alone
        # 9.21 -> 8.68 ms, the train step + 1.2 % -- the same + 1 % the persistent GEMM kernel (gemm_nt_pp_kernel) brings by removing the
        # per-launch costs at the source, and the two do not add (two half-chip persistent launches side by side lose).  The kernel route
        # ships; the chains stay available: DINOUNET_VIT_CHAINS=2, or `backbone.chains = 2`.  0 / 1 = one chain on the caller's stream.
        self.chains = int(os.environ.get("DINOUNET_VIT_CHAINS", "1"))
        self.overlap_prior = ops._ab_env("DINOUNET_VIT_OVERLAP_PRIOR", "1") != "0"     # the adapter's prior module beside the chains
        self._chain_streams = {}
        self._chain_ws = {}
        self.init_weights()

    def init_weights(self):
        """vision_transformer.py:178-184 + init_weights_vit :40-52; bias_mask set to 1|0|1 (a fresh reference model
        leaves it NaN, layers/attention.py:36 -- released checkpoints carry the mask)."""
        nn.init.normal_(self.cls_token, std=0.02)
        if self.n_storage_tokens > 0:
            nn.init.normal_(self.storage_tokens, std=0.02)
        nn.init.zeros_(self.mask_token)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
                if hasattr(m, "bias_mask"):
                    o = m.out_features // 3
                    m.bias_mask.fill_(1.0)
                    m.bias_mask[o:2 * o] = 0.0
        k = 1 / (3 * self.patch_size ** 2)
        nn.init.uniform_(self.patch_embed.proj.weight, -math.sqrt(k), math.sqrt(k))
        nn.init.uniform_(self.patch_embed.proj.bias, -math.sqrt(k), math.sqrt(k))

    # ------------------------------------------------------------------------------------------
    def _packed(self, dt, device):
        """Weights in the GEMM dtype, packed once (the backbone is frozen); rebuilt if any parameter changes."""
        ver = sum(p._version for p in self.parameters()) + sum(b._version for b in self.buffers())
        key = (dt, str(device), ver)
        if self._cache is not None and self._cache[0] == key:
            return self._cache[1]
        f = lambda t: t.detach().to(device=device, dtype=torch.float32).contiguous()
        g = lambda t: t.detach().to(device=device, dtype=dt).contiguous()
        pk = {"pe_w": g(self.patch_embed.proj.weight.flatten(1)), "pe_b": f(self.patch_embed.proj.bias),
              "norm_w": f(self.norm.weight), "norm_b": f(self.norm.bias), "blocks": []}
        prefix = self.cls_token + 0 * self.mask_token                                   # vision_transformer.py:195
        if self.n_storage_tokens > 0:
            prefix = torch.cat([prefix, self.storage_tokens], dim=1)
        pk["prefix"] = f(prefix[0])
        for blk in self.blocks:
            a = blk.attn
            qb = None
            if a.qkv.bias is not None:
                qb = a.qkv.bias.detach()
                if hasattr(a.qkv, "bias_mask"):
                    qb = qb * a.qkv.bias_mask.to(qb.dtype)
                qb = f(qb)
            d = dict(n1w=f(blk.norm1.weight), n1b=f(blk.norm1.bias), qkv_w=g(a.qkv.weight), qkv_b=qb,
                     proj_w=g(a.proj.weight), proj_b=f(a.proj.bias), g1=f(blk.ls1.gamma), n2w=f(blk.norm2.weight),
                     n2b=f(blk.norm2.bias), g2=f(blk.ls2.gamma))
            if isinstance(blk.mlp, Mlp):
                d.update(fc1_w=g(blk.mlp.fc1.weight), fc1_b=f(blk.mlp.fc1.bias), fc2_w=g(blk.mlp.fc2.weight), fc2_b=f(blk.mlp.fc2.bias))
            else:   # SwiGLU: w1 / w2 interleaved row by row -> one product whose epilogue applies the gate (ops.mm_swiglu)
                d.update(w12=g(ops.interleave_pairs(blk.mlp.w1.weight.detach(), blk.mlp.w2.weight.detach())),
                         b12=f(ops.interleave_pairs(blk.mlp.w1.bias.detach(), blk.mlp.w2.bias.detach())),
                         w3=g(blk.mlp.w3.weight), b3=f(blk.mlp.w3.bias))
            pk["blocks"].append(d)
        self._cache = (key, pk)
        return pk

    @torch.no_grad()
    def get_intermediate_layers(self, x, *, n, return_class_token=True, dtype=torch.bfloat16, norm=True):
        """x: (B, 3, H, W) fp32 NCHW on the GPU.  Returns [(patch tokens (B, h*w, D) in `dtype`, cls (B, D))] for the
        block indices in `n` (vision_transformer.py:265-318)."""
        return self.begin_intermediate_layers(x, n=n, return_class_token=return_class_token, dtype=dtype, norm=norm)()

    @torch.no_grad()
    def begin_intermediate_layers(self, x, *, n, return_class_token=True, dtype=torch.bfloat16, norm=True):
        """get_intermediate_layers in two halves: this call LAUNCHES the backbone (with `chains` > 1: as half-batch chains on side streams,
        forked from the caller's stream) and returns a function that joins the chains back into the caller's stream and returns the
        outputs -- the caller (DINOv3_Adapter) runs its spatial prior module, which does not depend on the ViT (ADP:412-426), in between."""
        B, _, H, W = x.shape
        D, nh = self.embed_dim, self.num_heads
        dh = D // nh
        hp, wp = H // self.patch_size, W // self.patch_size
        npre = 1 + self.n_storage_tokens
        N = npre + hp * wp
        pk = self._packed(dtype, x.device)
        cols = ops.patchify16(x, dtype)
        xs = torch.empty((B, N, D), dtype=torch.float32, device=x.device)                  # fp32 residual stream
        xs[:, :npre] = pk["prefix"]
        tok = ops.mm(cols, pk["pe_w"], bias=pk["pe_b"], out_dtype=torch.float32)
        xs[:, npre:] = tok.view(B, hp * wp, D)
        take = list(n)
        sin_all, cos_all = self.rope_embed.sincos_all(hp, wp, x.device, self.training, len(pk["blocks"]))   # vision_transformer.py:271-272
        rope_grid = (hp, wp)       # the tables are those of an hp x wp token grid: ops.qkv_attention may apply them from a factorised table
        sd = self.training and self.drop_path_rate > 0.0
        k_sub = max(int(B * (1 - self.drop_path_rate)), 1)                                 # layers/block.py:92-93
        if sd:
            alpha = torch.full((1,), B / k_sub, dtype=torch.float32, device=x.device)        # residual_scale_factor

        def attn_branch(xr, Bs, i, d, scale, ws):
            """xr (Bs*N, D) fp32 += [scale *] ls1(attn(norm1(xr)))   in place (layers/block.py:189-193)"""
            h, _, _ = ops.layernorm_raw(xr, d["n1w"], d["n1b"], 1e-5, dtype)
            a = ops.qkv_attention(h, d["qkv_w"], d["qkv_b"], sin_all[i], cos_all[i], Bs, N, nh, dh, npre, ws, grid=rope_grid)
            ops.mm(a, d["proj_w"], bias=d["proj_b"], gamma=d["g1"], residual=xr, out=xr, row_scale=scale, rs_rows=xr.shape[0] if scale is not None else 0)

        def ffn_branch(xr, d, scale):
            h, _, _ = ops.layernorm_raw(xr, d["n2w"], d["n2b"], 1e-5, dtype)
            rs = xr.shape[0] if scale is not None else 0
            if "fc1_w" in d:
                u = ops.mm(h, d["fc1_w"], bias=d["fc1_b"], act=ACT_GELU)
                ops.mm(u, d["fc2_w"], bias=d["fc2_b"], gamma=d["g2"], residual=xr, out=xr, row_scale=scale, rs_rows=rs)
            else:                                                                          # SwiGLU, ffn_layers.py:73-77
                u = ops.mm_swiglu(h, d["w12"], d["b12"])
                ops.mm(u, d["w3"], bias=d["b3"], gamma=d["g2"], residual=xr, out=xr, row_scale=scale, rs_rows=rs)

        # tap outputs: whole-batch buffers owned by the caller's stream, every chain writes its own samples' rows
        taps = [torch.empty((B, N, D), dtype=dtype, device=x.device) for _ in take]

        def run_chain(b0, b1, ws):
            """samples [b0, b1) through all blocks on the current stream"""
            Bs = b1 - b0
            xc = xs[b0:b1].view(Bs * N, D)
            t = 0
            for i, d in enumerate(pk["blocks"]):
                attn_branch(xc, Bs, i, d, None, ws)
                ffn_branch(xc, d, None)
                if i in take:
                    o = taps[t][b0:b1].view(Bs * N, D)
                    if norm:
                        ops.layernorm_raw(xc, pk["norm_w"], pk["norm_b"], 1e-5, dtype, out=o)   # vision_transformer.py:300
                    else:
                        o.copy_(xc)
                    t += 1
            assert t == len(take), f"only {t} / {len(take)} blocks found"

        nch = self.chains if (x.is_cuda and not sd and ops.PROFILE is None and self.chains > 1 and B >= 2 * self.chains
                              and dtype == torch.bfloat16) else 1
        pending = []
        if sd:
            # batch-subset stochastic depth (layers/block.py:126-187): each branch runs on a random subset of k_sub samples and its
            # residual is added back scaled by B / k_sub; the subset is gathered, updated in place and scattered back.  One chain: the
            # subsets mix the samples.
            x2 = xs.view(B * N, D)
            t = 0
            for i, d in enumerate(pk["blocks"]):
                if self.pinned_subsets is not None:
                    i1, i2 = (t_.to(x.device) for t_ in self.pinned_subsets[i])
                else:
                    i1 = torch.randperm(B, device=x.device)[:k_sub]
                    i2 = torch.randperm(B, device=x.device)[:k_sub]
                xsub = ops.sample_gather(xs, i1)
                attn_branch(xsub.view(k_sub * N, D), k_sub, i, d, alpha, self._ws)
                ops.sample_scatter_(xs, xsub, i1)
                xsub = ops.sample_gather(xs, i2)
                ffn_branch(xsub.view(k_sub * N, D), d, alpha)
                ops.sample_scatter_(xs, xsub, i2)
                if i in take:
                    o = taps[t].view(B * N, D)
                    if norm:
                        ops.layernorm_raw(x2, pk["norm_w"], pk["norm_b"], 1e-5, dtype, out=o)   # vision_transformer.py:300
                    else:
                        o.copy_(x2)
                    t += 1
            assert t == len(take), f"only {t} / {len(take)} blocks found"
        elif nch == 1:
            run_chain(0, B, self._ws)
        else:
            main = torch.cuda.current_stream()
            key = str(x.device)
            if key not in self._chain_streams:
                self._chain_streams[key] = [torch.cuda.Stream(device=x.device) for _ in range(8)]
            per = (B + nch - 1) // nch
            ops.set_corun(nch)                      # du_gemm's tile choice: each product shares the chip with nch - 1 twins
            try:
                for c in range(nch):
                    b0, b1 = c * per, min(B, (c + 1) * per)
                    if b0 >= b1:
                        continue
                    st = self._chain_streams[key][c]
                    st.wait_stream(main)                # fork: the chain starts behind patch embedding / the RoPE tables
                    with torch.cuda.stream(st):
                        run_chain(b0, b1, self._chain_ws.setdefault((key, c), {}))
                    pending.append(st)
            finally:
                ops.set_corun(1)

        # everything the chains read or update that belongs to the caller's stream stays referenced until the join: freed earlier, the caching
        # allocator would hand the blocks to the caller's next allocations (the spatial prior module) while the chains still use them
        keep = [xs, sin_all, cos_all, cols, tok]

        def join():
            if pending:
                cur = torch.cuda.current_stream()
                for st in pending:
                    cur.wait_stream(st)
                pending.clear()
            keep.clear()
            outs = [(o[:, npre:].contiguous(), o[:, 0].contiguous()) for o in taps]
            return tuple(outs) if return_class_token else tuple(o for o, _ in outs)

        return join


def build_backbone(model_name: str) -> DinoVisionTransformer:
    return DinoVisionTransformer(**VIT_CONFIGS[model_name])
