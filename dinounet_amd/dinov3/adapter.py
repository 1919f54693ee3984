"""DINOv3 ViT-Adapter (spatial prior module + deformable-attention extractors) on the HIP kernels.

Module tree / state_dict / initialisation mirror the reference
(dinounet/dinov3/eval/segmentation/models/backbone/dinov3_adapter.py, ADP below; MSDeformAttn from
.../utils/ms_deform_attn.py, MSA below).  torch sub-modules are parameter containers; forward and backward run in
libdinounet_hip.so through dinounet_amd.ops.  Activations are NHWC / token-major in the activation dtype.

Deviations that do not change results: activation checkpointing of each Extractor (ADP:151) is dropped (288 GB HBM);
sampling_offsets and attention_weights share one GEMM (their weights are concatenated per call); level_embed is
folded into the SPM fc biases.
"""
import math
from functools import partial

import torch
from torch import nn

from .. import ops
from .._lib import ACT_GELU, ACT_NONE, ACT_RELU


def _act_dtype(module):
    return getattr(module, "_act_dtype", torch.bfloat16)


class MSDeformAttn(nn.Module):
    """MSA:101-216.  Single call site here: n_levels=1, 16 heads, 4 points, ratio 0.5 (dinounet_training.py:759-765)."""

    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4, ratio=1.0):
        super().__init__()
        if d_model % n_heads != 0:
            raise ValueError("d_model must be divisible by n_heads, but got {} and {}".format(d_model, n_heads))
        self.im2col_step = 64
        self.d_model, self.n_levels, self.n_heads, self.n_points, self.ratio = d_model, n_levels, n_heads, n_points, ratio
        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = nn.Linear(d_model, int(d_model * ratio))
        self.output_proj = nn.Linear(int(d_model * ratio), d_model)
        self._reset_parameters()

    def _reset_parameters(self):
        """MSA:137-156."""
        nn.init.constant_(self.sampling_offsets.weight.data, 0.0)
        thetas = torch.arange(self.n_heads, dtype=torch.float32) * (2.0 * math.pi / self.n_heads)
        grid_init = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid_init = (grid_init / grid_init.abs().max(-1, keepdim=True)[0]).view(self.n_heads, 1, 1, 2).repeat(
            1, self.n_levels, self.n_points, 1)
        for i in range(self.n_points):
            grid_init[:, :, i, :] *= i + 1
        with torch.no_grad():
            self.sampling_offsets.bias = nn.Parameter(grid_init.view(-1))
        nn.init.constant_(self.attention_weights.weight.data, 0.0)
        nn.init.constant_(self.attention_weights.bias.data, 0.0)
        nn.init.xavier_uniform_(self.value_proj.weight.data)
        nn.init.constant_(self.value_proj.bias.data, 0.0)
        nn.init.xavier_uniform_(self.output_proj.weight.data)
        nn.init.constant_(self.output_proj.bias.data, 0.0)

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
                input_padding_mask=None, residual=None):
        """query (N, Lq, C), reference_points (Lq, 2) [(x, y), shared by the batch], input_flatten (N, S, C),
        input_spatial_shapes [(H, W)] python list (single level).  Returns output_proj(msda) (+ residual)."""
        assert input_padding_mask is None and self.n_levels == 1 and len(input_spatial_shapes) == 1
        N, Lq, _ = query.shape
        _, S, _ = input_flatten.shape
        Hs, Ws = input_spatial_shapes[0]
        assert Hs * Ws == S
        M, P = self.n_heads, self.n_points
        value = ops.linear(input_flatten, self.value_proj.weight, self.value_proj.bias)
        value = value.view(N, S, M, value.shape[-1] // M)
        # MSA:188-197: offsets | weights product (fp32, MSA:30) + reference-point / softmax step as one autograd node (ops._OffsetsPrep)
        loc, attn = ops.offsets_prep(query, self.sampling_offsets.weight, self.attention_weights.weight, self.sampling_offsets.bias,
                                     self.attention_weights.bias, reference_points, Lq, M, P, Hs, Ws)
        shapes, lsi = _level_tensors(Hs, Ws, query.device)   # cached: no host->device copy (sync) per call
        out = ops.msda(value, shapes, lsi, loc.view(N, Lq, M, 1, P, 2), attn.view(N, Lq, M, 1, P))   # MSA:207-214
        return ops.linear(out, self.output_proj.weight, self.output_proj.bias, residual=residual)


_LEVEL_CACHE = {}


def _level_tensors(Hs, Ws, device):
    key = (Hs, Ws, str(device))
    if key not in _LEVEL_CACHE:
        _LEVEL_CACHE[key] = (torch.tensor([[Hs, Ws]], dtype=torch.long, device=device),
                             torch.zeros(1, dtype=torch.long, device=device))
    return _LEVEL_CACHE[key]


class DWConv(nn.Module):
    def __init__(self, dim=768):
        super().__init__()
        self.dwconv = nn.Conv2d(dim, dim, 3, 1, 1, bias=True, groups=dim)


class ConvFFN(nn.Module):
    """ADP:73-91."""

    def __init__(self, in_features, hidden_features=None, out_features=None, drop=0.0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.dwconv = DWConv(hidden_features)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x, H, W, residual=None, row_scale=None):
        h = ops.linear(x, self.fc1.weight, self.fc1.bias)
        h = ops.dwconv_tokens(h, self.dwconv.dwconv.weight, self.dwconv.dwconv.bias, H, W, ACT_GELU)   # ADP:99-109 + :87
        return ops.linear(h, self.fc2.weight, self.fc2.bias, residual=residual, row_scale=row_scale, rs_rows=x.shape[1])


class DropPath(nn.Module):
    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = drop_prob
        self.pinned_mask = None             # test hook: a fixed per-sample mask instead of the device RNG draw
        self.predrawn = None                # this forward's mask, drawn for all DropPath modules at once by DINOv3_Adapter.forward

    def mask(self, B, device):
        """ADP:18-26: per-sample Bernoulli(keep)/keep, or None when inactive."""
        if self.drop_prob == 0.0 or not self.training:
            return None
        if self.pinned_mask is not None:            # parity tests feed the mask the reference was given (already / keep_prob)
            return self.pinned_mask.to(device=device, dtype=torch.float32)
        if self.predrawn is not None and self.predrawn.shape[0] == B:
            m, self.predrawn = self.predrawn, None
            return m
        keep = 1 - self.drop_prob
        m = torch.empty(B, device=device, dtype=torch.float32).bernoulli_(keep)
        if keep > 0.0:
            m.div_(keep)
        return m


class Extractor(nn.Module):
    """ADP:112-156."""

    def __init__(self, dim, num_heads=6, n_points=4, n_levels=1, deform_ratio=1.0, with_cffn=True, cffn_ratio=0.25, drop=0.0,
                 drop_path=0.0, norm_layer=partial(nn.LayerNorm, eps=1e-6), with_cp=False):
        super().__init__()
        self.query_norm = norm_layer(dim)
        self.feat_norm = norm_layer(dim)
        self.attn = MSDeformAttn(d_model=dim, n_levels=n_levels, n_heads=num_heads, n_points=n_points, ratio=deform_ratio)
        self.with_cffn = with_cffn
        self.with_cp = with_cp
        if with_cffn:
            self.ffn = ConvFFN(in_features=dim, hidden_features=int(dim * cffn_ratio), drop=drop)
            self.ffn_norm = norm_layer(dim)
            self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()

    def forward(self, query, reference_points, feat, spatial_shapes, level_start_index, H, W):
        qn, qres = ops.layer_norm_res(query, self.query_norm.weight, self.query_norm.bias, self.query_norm.eps)
        fn = ops.layer_norm(feat, self.feat_norm.weight, self.feat_norm.bias, self.feat_norm.eps)
        query = self.attn(qn, reference_points, fn, spatial_shapes, level_start_index, None, residual=qres)    # ADP:142-145
        if self.with_cffn:
            f, fres = ops.layer_norm_res(query, self.ffn_norm.weight, self.ffn_norm.bias, self.ffn_norm.eps)
            mask = self.drop_path.mask(query.shape[0], query.device) if isinstance(self.drop_path, DropPath) else None
            query = self.ffn(f, H, W, residual=fres, row_scale=mask)                                             # ADP:148
        return query


class InteractionBlockWithCls(nn.Module):
    """ADP:159-231."""

    def __init__(self, dim, num_heads=6, n_points=4, norm_layer=partial(nn.LayerNorm, eps=1e-6), drop=0.0, drop_path=0.0,
                 with_cffn=True, cffn_ratio=0.25, init_values=0.0, deform_ratio=1.0, extra_extractor=False, with_cp=False):
        super().__init__()
        mk = lambda: Extractor(dim=dim, n_levels=1, num_heads=num_heads, n_points=n_points, norm_layer=norm_layer,
                               deform_ratio=deform_ratio, with_cffn=with_cffn, cffn_ratio=cffn_ratio, drop=drop,
                               drop_path=drop_path, with_cp=with_cp)
        self.extractor = mk()
        self.extra_extractors = nn.Sequential(*[mk() for _ in range(2)]) if extra_extractor else None

    def forward(self, x, c, ref, shapes, H_c, W_c):
        c = self.extractor(c, ref, x, shapes, None, H_c, W_c)
        if self.extra_extractors is not None:
            for ex in self.extra_extractors:
                c = ex(c, ref, x, shapes, None, H_c, W_c)
        return c


class SpatialPriorModule(nn.Module):
    """ADP:234-302 (conv stem on NHWC; BatchNorm statistics are synchronised over the process group like SyncBatchNorm)."""

    def __init__(self, inplanes=64, embed_dim=384, with_cp=False):
        super().__init__()
        bn = nn.SyncBatchNorm
        self.stem = nn.Sequential(
            nn.Conv2d(3, inplanes, kernel_size=3, stride=2, padding=1, bias=False), bn(inplanes), nn.ReLU(inplace=True),
            nn.Conv2d(inplanes, inplanes, kernel_size=3, stride=1, padding=1, bias=False), bn(inplanes), nn.ReLU(inplace=True),
            nn.Conv2d(inplanes, inplanes, kernel_size=3, stride=1, padding=1, bias=False), bn(inplanes), nn.ReLU(inplace=True),
            nn.MaxPool2d(kernel_size=3, stride=2, padding=1))
        self.conv2 = nn.Sequential(nn.Conv2d(inplanes, 2 * inplanes, 3, 2, 1, bias=False), bn(2 * inplanes), nn.ReLU(inplace=True))
        self.conv3 = nn.Sequential(nn.Conv2d(2 * inplanes, 4 * inplanes, 3, 2, 1, bias=False), bn(4 * inplanes), nn.ReLU(inplace=True))
        self.conv4 = nn.Sequential(nn.Conv2d(4 * inplanes, 4 * inplanes, 3, 2, 1, bias=False), bn(4 * inplanes), nn.ReLU(inplace=True))
        self.fc1 = nn.Conv2d(inplanes, embed_dim, kernel_size=1)
        self.fc2 = nn.Conv2d(2 * inplanes, embed_dim, kernel_size=1)
        self.fc3 = nn.Conv2d(4 * inplanes, embed_dim, kernel_size=1)
        self.fc4 = nn.Conv2d(4 * inplanes, embed_dim, kernel_size=1)

    def _cbr(self, x, conv, bn, stride, group):
        w = conv.weight
        if w.shape[1] % 8:                                   # 3-channel stem: image is zero-padded to 8 channels
            w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, 8 - w.shape[1] % 8))
        y, st = ops.conv2d_stats(x, w, None, stride=stride, pad=1)
        return bn_act(y, bn, ACT_RELU, self.training, group, st)

    def forward(self, x8, level_embed, group=None):
        """x8: NHWC image zero-padded to 8 channels.  Returns c1 (B,H/4,W/4,D) and token tensors c2,c3,c4 with the level
        embedding (ADP:402-406) already added through the fc bias."""
        c = self._cbr(x8, self.stem[0], self.stem[1], 2, group)
        c = self._cbr(c, self.stem[3], self.stem[4], 1, group)
        c = self._cbr(c, self.stem[6], self.stem[7], 1, group)
        c1 = ops.maxpool3x3s2(c)
        c2 = self._cbr(c1, self.conv2[0], self.conv2[1], 2, group)
        c3 = self._cbr(c2, self.conv3[0], self.conv3[1], 2, group)
        c4 = self._cbr(c3, self.conv4[0], self.conv4[1], 2, group)
        c1 = ops.conv1x1(c1, self.fc1.weight, self.fc1.bias)
        c2 = ops.conv1x1(c2, self.fc2.weight, self.fc2.bias + level_embed[0])
        c3 = ops.conv1x1(c3, self.fc3.weight, self.fc3.bias + level_embed[1])
        c4 = ops.conv1x1(c4, self.fc4.weight, self.fc4.bias + level_embed[2])
        tok = lambda t: t.view(t.shape[0], -1, t.shape[-1])
        return c1, tok(c2), tok(c3), tok(c4)


_TICK_STACK = []     # one list per running DINOv3_Adapter.forward: the num_batches_tracked counters to bump when it ends


def _tick(bn):
    """nn.BatchNorm2d.forward's `num_batches_tracked += 1` (state_dict parity with the reference).  Inside DINOv3_Adapter.forward the
    bumps are collected and applied by ONE torch._foreach_add_ when the forward ends (ten scalar adds are ten ~5 us launches in the
    replayed step); a BatchNorm used on its own (SpatialPriorModule / bn_act outside an adapter forward) is bumped at once."""
    if _TICK_STACK:
        _TICK_STACK[-1].append(bn.num_batches_tracked)
    else:
        bn.num_batches_tracked += 1


class _collect_ticks:
    """context of one adapter forward: nested / concurrent adapters each own a list; the bumps of the BatchNorms that ran are applied
    even when the forward leaves through an exception (their running statistics were updated as well)"""

    def __enter__(self):
        _TICK_STACK.append([])
        return self

    def __exit__(self, *exc):
        ticks = _TICK_STACK.pop()
        if ticks:
            torch._foreach_add_(ticks, 1)
        return False


def bn_act(x, bn, act, training, group, stats_part=None):
    if training and bn.track_running_stats and bn.num_batches_tracked is not None:
        _tick(bn)
    return ops.norm_act(x, bn.weight, bn.bias, "bn", act, bn.eps, training, bn.running_mean, bn.running_var,
                        bn.momentum if bn.momentum is not None else 0.1, group, stats_part=stats_part if training else None)


def get_reference_points(spatial_shapes, device):
    """ADP:40-53 -> (sum HW, 2) fp32 (x, y)."""
    refs = []
    for (H_, W_) in spatial_shapes:
        ry, rx = torch.meshgrid(torch.linspace(0.5, H_ - 0.5, H_, dtype=torch.float32),
                                torch.linspace(0.5, W_ - 0.5, W_, dtype=torch.float32), indexing="ij")
        refs.append(torch.stack((rx.reshape(-1) / W_, ry.reshape(-1) / H_), -1))
    return torch.cat(refs, 0).to(device).contiguous()


class DINOv3_Adapter(nn.Module):
    """ADP:305-484."""

    def __init__(self, backbone, interaction_indexes=[9, 19, 29, 39], pretrain_size=512, conv_inplane=64, n_points=4,
                 deform_num_heads=16, drop_path_rate=0.3, init_values=0.0, with_cffn=True, cffn_ratio=0.25, deform_ratio=0.5,
                 add_vit_feature=True, use_extra_extractor=True, with_cp=True):
        super().__init__()
        self.backbone = backbone
        self.backbone.requires_grad_(False)                   # ADP:326
        self.pretrain_size = (pretrain_size, pretrain_size)
        self.interaction_indexes = interaction_indexes
        self.add_vit_feature = add_vit_feature
        embed_dim = self.backbone.embed_dim
        self.patch_size = self.backbone.patch_size
        self.level_embed = nn.Parameter(torch.zeros(3, embed_dim))
        self.spm = SpatialPriorModule(inplanes=conv_inplane, embed_dim=embed_dim, with_cp=False)
        self.interactions = nn.Sequential(*[
            InteractionBlockWithCls(dim=embed_dim, num_heads=deform_num_heads, n_points=n_points, init_values=init_values,
                                    drop_path=drop_path_rate, norm_layer=partial(nn.LayerNorm, eps=1e-6), with_cffn=with_cffn,
                                    cffn_ratio=cffn_ratio, deform_ratio=deform_ratio,
                                    extra_extractor=((i == len(interaction_indexes) - 1) and use_extra_extractor), with_cp=with_cp)
            for i in range(len(interaction_indexes))])
        self.up = nn.ConvTranspose2d(embed_dim, embed_dim, 2, 2)
        self.norm1 = nn.SyncBatchNorm(embed_dim)
        self.norm2 = nn.SyncBatchNorm(embed_dim)
        self.norm3 = nn.SyncBatchNorm(embed_dim)
        self.norm4 = nn.SyncBatchNorm(embed_dim)
        self.up.apply(self._init_weights)
        self.spm.apply(self._init_weights)
        self.interactions.apply(self._init_weights)
        self.apply(self._init_deform_weights)
        torch.nn.init.normal_(self.level_embed)
        self._ref_cache = {}

    def _init_weights(self, m):
        """ADP:372-385."""
        if isinstance(m, nn.Linear):
            torch.nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, (nn.LayerNorm, nn.BatchNorm2d)):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)
        elif isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
            fan_out //= m.groups
            m.weight.data.normal_(0, math.sqrt(2.0 / fan_out))
            if m.bias is not None:
                m.bias.data.zero_()

    def _init_deform_weights(self, m):
        if isinstance(m, MSDeformAttn):
            m._reset_parameters()

    def forward(self, x):
        """x: (B, 3, H, W) fp32 NCHW.  Returns {"1".."4"}: NHWC feature maps (B, H/4.., W/4.., D) in the activation dtype."""
        with _collect_ticks():
            return self._forward(x)

    def _forward(self, x):
        dt = _act_dtype(self)
        B, _, H, W = x.shape
        D = self.backbone.embed_dim
        H_c, W_c = H // 16, W // 16
        H_t, W_t = H // self.patch_size, W // self.patch_size
        group = torch.distributed.group.WORLD if (torch.distributed.is_available() and torch.distributed.is_initialized()) else None
        key = (H, W, str(x.device))
        if key not in self._ref_cache:
            self._ref_cache = {key: get_reference_points([(H // 8, W // 8), (H // 16, W // 16), (H // 32, W // 32)], x.device)}
        ref = self._ref_cache[key]                              # deform_inputs2, ADP:65-68
        shapes = [(H_t, W_t)]

        if self.training:
            # all DropPath masks of this forward in one draw (ADP:18-26 draws per call: same distribution, 2 launches instead of 12)
            dps = [m for m in self.modules() if isinstance(m, DropPath) and m.drop_prob > 0.0 and m.pinned_mask is None]
            keeps = {1 - m.drop_prob for m in dps}
            if len(dps) > 1 and len(keeps) == 1:
                keep = keeps.pop()
                tbl = torch.empty((len(dps), x.shape[0]), device=x.device, dtype=torch.float32).bernoulli_(keep)
                if keep > 0.0:
                    tbl.div_(keep)
                for i, m in enumerate(dps):
                    m.predrawn = tbl[i]
        # The frozen backbone is launched FIRST (ADP:422-426): with `backbone.chains` > 1 it runs as half-batch chains on side streams and the
        # spatial prior module below -- independent of it (ADP:412-415), HBM-bound where the ViT is MFMA-bound -- runs beside them on this
        # stream; `vit()` joins the chains.  (chains <= 1: the backbone simply runs before the prior module; same results either way.)
        chained = getattr(self.backbone, "chains", 1) > 1
        if chained:
            vit = self.backbone.begin_intermediate_layers(x, n=self.interaction_indexes, return_class_token=True, dtype=dt)
            if not getattr(self.backbone, "overlap_prior", True):
                vit_out = vit()                             # (A-B aid: join at once, the prior module runs behind the backbone)
                vit = lambda: vit_out
        x8 = ops.nchw_to_nhwc(x, dt, 8)
        c1, c2, c3, c4 = self.spm(x8, self.level_embed, group)                              # ADP:412-413
        n2, n3 = c2.shape[1], c3.shape[1]
        c = torch.cat([c2, c3, c4], dim=1)                                                   # ADP:415

        if chained:
            layers = vit()
        else:       # one chain on this stream: the reference's order (prior module, then the backbone, ADP:412-426)
            layers = self.backbone.get_intermediate_layers(x, n=self.interaction_indexes, return_class_token=True, dtype=dt)

        for i, layer in enumerate(self.interactions):                                       # ADP:444-457
            xi, _cls = layers[i]
            c = layer(xi, c, ref, shapes, H_c, W_c)

        # ADP:460-466.  split (not three slices): its backward is ONE concatenation of the three gradients instead of three
        # zero-filled full-size tensors, three slice copies and two adds
        c2, c3, c4 = c.split([n2, n3, c.shape[1] - n2 - n3], dim=1)
        c2 = c2.contiguous().view(B, H_c * 2, W_c * 2, D)
        c3 = c3.contiguous().view(B, H_c, W_c, D)
        c4 = c4.contiguous().view(B, H_c // 2, W_c // 2, D)
        c1 = ops.conv_transpose2x2(c2, self.up.weight, self.up.bias, residual=c1)          # ADP:467 (add fused in the epilogue)
        cs = [c1, c2, c3, c4]
        if self.add_vit_feature:                                                            # ADP:469-476
            cs = [ops.bilinear_add(layers[j][0].view(B, H_t, W_t, D), cs[j]) for j in range(4)]
        norms = [self.norm1, self.norm2, self.norm3, self.norm4]
        if (self.training and group is not None and ops.sync_active(group)
                and all(bn.track_running_stats for bn in norms)):
            for bn in norms:                                                                # ADP:479-482, one packed collective each way
                if bn.num_batches_tracked is not None:
                    _tick(bn)
            fs = ops.sync_bn_multi(cs, norms, ACT_NONE, group)
        else:
            fs = [bn_act(cs[j], norms[j], ACT_NONE, self.training, group) for j in range(4)]    # ADP:479-482
        return {"1": fs[0], "2": fs[1], "3": fs[2], "4": fs[3]}
