"""Dino U-Net on the MI355X HIP kernels: the classes of the reference's `dinounet_training.py` (DT below) with the
same constructor arguments, attribute surface and state_dict layout, whose forward/backward run in
libdinounet_hip.so.  nn.Conv2d / nn.InstanceNorm2d / ... sub-modules are parameter containers only.

Interface contract: (B, C, H, W) float NCHW in -> (B, K, H, W) fp32 logits out (list when deep
supervision is on); `.decoder.deep_supervision`; frozen backbone params have requires_grad=False; state_dict keys
including the `decoder.encoder.*` duplicates (DT:549) and the third-party `all_modules.N` aliases.
"""
import os
import pydoc
from typing import List, Tuple, Type, Union

import torch
from torch import nn

from .. import ops
from .._lib import ACT_LEAKY, ACT_NONE, ACT_RELU
from ..dinov3 import DINOv3_Adapter, build_backbone

DINOv3_MODEL_FACTORIES = {n: (lambda n=n, **kw: build_backbone(n)) for n in ("dinounet_s", "dinounet_b", "dinounet_l", "dinounet_7b")}
DINOv3_INTERACTION_INDEXES = {"dinounet_s": [2, 5, 8, 11], "dinounet_b": [2, 5, 8, 11], "dinounet_l": [4, 11, 17, 23],
                              "dinounet_7b": [9, 19, 29, 39]}                                   # DT:36-41
DINOv3_MODEL_INFO = {                                                                             # DT:43-48
    "dinounet_s": {"embed_dim": 384, "depth": 12, "num_heads": 6, "params": "~22M"},
    "dinounet_b": {"embed_dim": 768, "depth": 12, "num_heads": 12, "params": "~86M"},
    "dinounet_l": {"embed_dim": 1024, "depth": 24, "num_heads": 16, "params": "~300M"},
    "dinounet_7b": {"embed_dim": 4096, "depth": 40, "num_heads": 32, "params": "~7B"},
}


def load_dinov3_model(model_name: str, pretrained_path: str = None, allow_random_backbone: bool = None):
    """DT:51-75.  With a checkpoint path the upstream DINOv3 state_dict is loaded strict=True.  The reference falls back to the hub
    download when the path is missing (`model_factory(pretrained=True)`, hub/backbones.py:140); there is no network here, and a
    training run on a randomly initialised FROZEN backbone is silently useless, so:
      * a non-None path that does not exist raises FileNotFoundError;
      * no path at all keeps the seeded random init only for callers that ask for it (`allow_random_backbone=True`, or the
        environment variable DINOUNET_ALLOW_RANDOM_BACKBONE=1 that tests / bench.py / smoke() set) and warns loudly otherwise."""
    if model_name not in DINOv3_MODEL_FACTORIES:
        raise ValueError(f"Unsupported model: {model_name}. Supported models: {list(DINOv3_MODEL_FACTORIES.keys())}")
    model = build_backbone(model_name)
    if pretrained_path:
        if not os.path.exists(pretrained_path):
            raise FileNotFoundError(f"DINOv3 checkpoint {pretrained_path!r} not found (the frozen backbone would stay randomly "
                                    f"initialised; the reference would download the default weights here, which this offline "
                                    f"build cannot)")
        state_dict = torch.load(pretrained_path, map_location="cpu")
        model.load_state_dict(state_dict, strict=True)
        return model
    if allow_random_backbone is None:
        allow_random_backbone = os.environ.get("DINOUNET_ALLOW_RANDOM_BACKBONE") == "1"
    if not allow_random_backbone:
        import warnings
        warnings.warn(f"load_dinov3_model({model_name!r}): no pretrained_path -- the FROZEN DINOv3 backbone keeps its random "
                      f"initialisation (fine for parity tests and synthetic benchmarks, useless for training); pass "
                      f"allow_random_backbone=True / set DINOUNET_ALLOW_RANDOM_BACKBONE=1 to silence this", RuntimeWarning, stacklevel=2)
    return model


def _act_code(nonlin, nonlin_kwargs):
    """Activation code of the fused norm + activation kernel, or None when the kernel has no such activation (the block then applies the
    nn.Module itself after an un-activated norm: a LeakyReLU slope other than 0.01, GELU, PReLU ... -- never hit by the 2D plans, but the
    reference's constructor accepts them, dinounet_training.py:581-592)."""
    if nonlin is None:
        return ACT_NONE
    if issubclass(nonlin, nn.LeakyReLU):
        slope = (nonlin_kwargs or {}).get("negative_slope", 0.01)
        return ACT_LEAKY if abs(slope - 0.01) <= 1e-12 else None
    if issubclass(nonlin, nn.ReLU):
        return ACT_RELU
    return None


def _nchw_module(mod, x):
    """Run a torch module that expects NCHW on our NHWC tensor (a channels-last view: no copy)."""
    return mod(x.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)


def _norm_act(x, norm_mod, act, training, stats_part=None):
    """InstanceNorm2d (plans default) or BatchNorm2d container -> fused norm+activation kernel.  stats_part: partial channel
    statistics emitted by the producing convolution (ops.conv2d_stats), saves the statistics pass."""
    if isinstance(norm_mod, nn.InstanceNorm2d) and not norm_mod.track_running_stats:
        w, b = norm_mod.weight, norm_mod.bias
        if not norm_mod.affine:                       # the kernel wants a scale / shift: ones / zeros (their gradients are dropped)
            w = torch.ones(x.shape[-1], dtype=torch.float32, device=x.device)
            b = torch.zeros(x.shape[-1], dtype=torch.float32, device=x.device)
        return ops.norm_act(x, w, b, "in", act, norm_mod.eps, True, stats_part=stats_part)
    if isinstance(norm_mod, nn.modules.batchnorm._BatchNorm):
        if training and norm_mod.track_running_stats and norm_mod.num_batches_tracked is not None:
            norm_mod.num_batches_tracked += 1          # nn.BatchNorm2d.forward does (state_dict parity with the reference)
        return ops.norm_act(x, norm_mod.weight, norm_mod.bias, "bn", act, norm_mod.eps, training, norm_mod.running_mean,
                            norm_mod.running_var, norm_mod.momentum or 0.1, None, stats_part=stats_part if training else None)
    raise TypeError(f"_norm_act: no kernel for {type(norm_mod)} (ConvDropoutNormReLU routes such layers through the torch module)")


class SqueezeExcitation(nn.Module):
    """DT:210-225."""

    def __init__(self, channels: int, reduction: int = 16):
        super().__init__()
        reduced = max(1, channels // reduction)
        self.pool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Sequential(nn.Conv2d(channels, reduced, kernel_size=1, bias=True), nn.ReLU(inplace=True),
                                nn.Conv2d(reduced, channels, kernel_size=1, bias=True), nn.Sigmoid())

    def forward(self, x, shortcut=None):
        """x NHWC; the residual add of DT:438 is fused into the scaling kernel."""
        return ops.squeeze_excite(x, self.fc[0].weight, self.fc[0].bias, self.fc[2].weight, self.fc[2].bias, shortcut)


class DepthwiseSeparableConv(nn.Module):
    """DT:228-246."""

    def __init__(self, in_ch, out_ch, kernel_size=3, stride=1, padding=1, bias=False, norm=nn.BatchNorm2d, act=nn.ReLU,
                 norm_kwargs=None, act_kwargs=None):
        super().__init__()
        norm_kwargs = {} if norm_kwargs is None else norm_kwargs
        act_kwargs = {"inplace": True} if act_kwargs is None else act_kwargs
        assert kernel_size == 3 and stride == 1 and padding == 1
        self.depthwise = nn.Conv2d(in_ch, in_ch, kernel_size=kernel_size, stride=stride, padding=padding, groups=in_ch, bias=bias)
        self.pointwise = nn.Conv2d(in_ch, out_ch, kernel_size=1, bias=bias)
        self.bn = norm(out_ch, **norm_kwargs) if norm is not None else nn.Identity()
        self.act = act(**act_kwargs) if act is not None else nn.Identity()
        self._act = _act_code(act, act_kwargs)

    def forward(self, x):
        x = ops.dwconv3x3(x, self.depthwise.weight, self.depthwise.bias)
        x = ops.conv1x1(x, self.pointwise.weight, self.pointwise.bias)
        return _norm_act(x, self.bn, self._act, self.training)


class LearnableUpsampleBlock(nn.Module):
    """DT:249-264: the same ConvTranspose2d(k2,s2) applied while a doubling still fits, then a bilinear resize to the exact target."""

    def __init__(self, channels: int):
        super().__init__()
        self.up2 = nn.ConvTranspose2d(channels, channels, kernel_size=2, stride=2, bias=True)

    def forward(self, x, target_size: Tuple[int, int]):
        h, w = x.shape[1], x.shape[2]
        out = x
        while h * 2 <= target_size[0] and w * 2 <= target_size[1]:
            out = ops.conv_transpose2x2(out, self.up2.weight, self.up2.bias)
            h, w = out.shape[1], out.shape[2]
        if (h, w) != tuple(target_size):                     # DT:262-263: final bilinear resize to the exact size
            out = ops.bilinear_resize(out, target_size)
        return out


class FAPM(nn.Module):
    """Feature Adaptive Projection Module, DT:355-441."""

    def __init__(self, in_ch, rank, out_ch_list, norm=nn.BatchNorm2d, act=nn.ReLU, norm_kwargs=None, act_kwargs=None, bias=False):
        super().__init__()
        norm_kwargs = {} if norm_kwargs is None else norm_kwargs
        act_kwargs = {"inplace": True} if act_kwargs is None else act_kwargs
        self.shared_basis = nn.Conv2d(in_ch, rank, kernel_size=1, bias=bias)
        self.specific_bases = nn.ModuleList([nn.Conv2d(in_ch, rank, kernel_size=1, bias=bias) for _ in out_ch_list])
        self.film_generators = nn.ModuleList([nn.Conv2d(rank, rank * 2, kernel_size=1, bias=bias) for _ in out_ch_list])
        self.refinement_blocks = nn.ModuleList()
        self.shortcut_projections = nn.ModuleList()
        self._act = _act_code(act, act_kwargs)
        for oc in out_ch_list:
            reduce = nn.Conv2d(rank, oc, kernel_size=1, bias=bias)
            dw = DepthwiseSeparableConv(oc, oc, 3, 1, 1, bias=bias, norm=norm, act=act, norm_kwargs=norm_kwargs, act_kwargs=act_kwargs)
            refine = nn.Conv2d(oc, oc, kernel_size=1, bias=bias)
            se = SqueezeExcitation(oc)
            self.refinement_blocks.append(nn.Sequential(
                reduce, norm(oc, **norm_kwargs) if norm is not None else nn.Identity(),
                act(**act_kwargs) if act is not None else nn.Identity(), dw, refine, se))
            self.shortcut_projections.append(nn.Conv2d(rank, oc, kernel_size=1, bias=bias) if rank != oc else nn.Identity())

    def forward(self, x_list):
        out = []
        rank = self.shared_basis.out_channels
        for i, x in enumerate(x_list):
            # shared + specific bases read the same D-channel input: one GEMM with 2*rank output columns (DT:423-424)
            sb, sp = self.shared_basis, self.specific_bases[i]
            fg = self.film_generators[i]
            z = ops.fapm_project(x, sb.weight, sp.weight, sb.bias, sp.bias, fg.weight, fg.bias)      # DT:423-429, one autograd node
            r = self.refinement_blocks[i]
            t = ops.conv1x1(z, r[0].weight, r[0].bias)
            t = _norm_act(t, r[1], self._act, self.training)
            t = r[3](t)
            t = ops.conv1x1(t, r[4].weight, r[4].bias)
            sc = self.shortcut_projections[i]
            short = z if isinstance(sc, nn.Identity) else ops.conv1x1(z, sc.weight, sc.bias)  # DT:436
            out.append(r[5](t, short))                                                         # SE + residual, DT:438
        return out


class DINOv3EncoderAdapter(nn.Module):
    """DT:444-514."""

    def __init__(self, dinov3_adapter, target_channels, adapter_type="default", rank=256, conv_op=nn.Conv2d,
                 norm_op=nn.BatchNorm2d, norm_op_kwargs=None, dropout_op=None, dropout_op_kwargs=None, nonlin=nn.ReLU,
                 nonlin_kwargs=None, conv_bias=False):
        super().__init__()
        self.dinov3_adapter = dinov3_adapter
        self.target_channels = target_channels
        self.conv_op = conv_op
        self.norm_op = norm_op if norm_op is not None else nn.BatchNorm2d
        self.norm_op_kwargs = norm_op_kwargs if norm_op_kwargs is not None else {}
        self.nonlin = nonlin if nonlin is not None else nn.ReLU
        self.nonlin_kwargs = nonlin_kwargs if nonlin_kwargs is not None else {"inplace": True}
        self.conv_bias = conv_bias
        self.dropout_op = dropout_op
        self.dropout_op_kwargs = dropout_op_kwargs
        in_ch = self.dinov3_adapter.backbone.embed_dim
        self.fapm = FAPM(in_ch, rank, target_channels, norm=self.norm_op, act=self.nonlin, norm_kwargs=self.norm_op_kwargs,
                         act_kwargs=self.nonlin_kwargs, bias=conv_bias)
        self.ups = nn.ModuleList([LearnableUpsampleBlock(oc) for oc in target_channels])
        self.output_channels = target_channels
        self.strides = [[2, 2]] * len(target_channels)
        self.kernel_sizes = [[3, 3]] * len(target_channels)

    def forward(self, x):
        """x (B, C, H, W) fp32 NCHW -> 4 NHWC skips (DT:489-511)."""
        B, C, H, W = x.shape
        if C == 1:
            x = x.repeat(1, 3, 1, 1)
        elif C != 3:
            x = x.repeat(1, 3 // C + (1 if 3 % C != 0 else 0), 1, 1)[:, :3] if C < 3 else x[:, :3]
        feats = self.dinov3_adapter(x)
        ys = self.fapm([feats[k] for k in ("1", "2", "3", "4")])
        return [self.ups[i](y, (H // (2 ** i), W // (2 ** i))) for i, y in enumerate(ys)]

    def compute_conv_feature_map_size(self, input_size):
        return 0


class ConvDropoutNormReLU(nn.Module):
    """Stand-in for dynamic_network_architectures.building_blocks.simple_conv_blocks.ConvDropoutNormReLU (0.4.x; the
    package is not vendored in the reference -- requirements.txt:3): conv(k, pad=(k-1)//2, bias) -> norm -> nonlin, with
    the `all_modules` nn.Sequential that re-registers the same sub-modules (hence the aliased state_dict keys)."""

    def __init__(self, conv_op, input_channels, output_channels, kernel_size, stride, conv_bias=False, norm_op=None,
                 norm_op_kwargs=None, dropout_op=None, dropout_op_kwargs=None, nonlin=None, nonlin_kwargs=None, nonlin_first=False):
        super().__init__()
        if not isinstance(kernel_size, (tuple, list)):
            kernel_size = [kernel_size] * 2
        if not isinstance(stride, (tuple, list)):
            stride = [stride] * 2
        mods = []
        self.conv = conv_op(input_channels, output_channels, kernel_size, stride, padding=[(i - 1) // 2 for i in kernel_size],
                            dilation=1, bias=conv_bias)
        mods.append(self.conv)
        if dropout_op is not None:
            self.dropout = dropout_op(**(dropout_op_kwargs or {}))
            mods.append(self.dropout)
        if norm_op is not None:
            self.norm = norm_op(output_channels, **(norm_op_kwargs or {}))
            mods.append(self.norm)
        if nonlin is not None:
            self.nonlin = nonlin(**(nonlin_kwargs or {}))
            mods.append(self.nonlin)
        if nonlin_first and norm_op is not None and nonlin is not None:     # conv -> dropout -> nonlin -> norm
            mods[-1], mods[-2] = mods[-2], mods[-1]
        self.nonlin_first = bool(nonlin_first)
        self._act = _act_code(nonlin, nonlin_kwargs)
        self.all_modules = nn.Sequential(*mods)

    def forward(self, x, x2=None):
        k, s = self.conv.kernel_size[0], self.conv.stride[0]
        y, st = ops.conv2d_stats(x, self.conv.weight, self.conv.bias, stride=s, pad=(k - 1) // 2, x2=x2)
        kernel_norm = hasattr(self, "norm") and (isinstance(self.norm, nn.modules.batchnorm._BatchNorm) or
                                                 (isinstance(self.norm, nn.InstanceNorm2d) and not self.norm.track_running_stats))
        fused = kernel_norm and not hasattr(self, "dropout") and not self.nonlin_first and self._act is not None
        if fused:                                     # what the 2D plans build: conv -> norm + activation in one kernel
            return _norm_act(y, self.norm, self._act, self.training, st)
        # the other orders / layers the reference's constructor accepts (dinounet_training.py:581-592), composed from the same pieces; the
        # convolution's own statistics describe its raw output only, so they are used only when the norm directly follows it
        if hasattr(self, "dropout"):
            y, st = _nchw_module(self.dropout, y), None
        has_act = hasattr(self, "nonlin")
        if self.nonlin_first and has_act:
            y, st = _nchw_module(self.nonlin, y), None
        late = has_act and not self.nonlin_first                # the activation still to come after the norm
        if kernel_norm:
            y = _norm_act(y, self.norm, self._act if late and self._act is not None else ACT_NONE, self.training, st)
            late = late and self._act is None
        elif hasattr(self, "norm"):                             # GroupNorm, InstanceNorm2d with running statistics ...: the torch module itself
            y = _nchw_module(self.norm, y)
        return _nchw_module(self.nonlin, y) if late else y


class StackedConvBlocks(nn.Module):
    def __init__(self, num_convs, conv_op, input_channels, output_channels, kernel_size, initial_stride, conv_bias=False,
                 norm_op=None, norm_op_kwargs=None, dropout_op=None, dropout_op_kwargs=None, nonlin=None, nonlin_kwargs=None,
                 nonlin_first=False):
        super().__init__()
        if not isinstance(output_channels, (tuple, list)):
            output_channels = [output_channels] * num_convs
        mk = lambda ci, co, st: ConvDropoutNormReLU(conv_op, ci, co, kernel_size, st, conv_bias, norm_op, norm_op_kwargs, dropout_op,
                                                    dropout_op_kwargs, nonlin, nonlin_kwargs, nonlin_first)
        self.convs = nn.Sequential(mk(input_channels, output_channels[0], initial_stride),
                                   *[mk(output_channels[i - 1], output_channels[i], 1) for i in range(1, num_convs)])

    def forward(self, x, x2=None):
        for i, blk in enumerate(self.convs):
            x = blk(x, x2) if i == 0 else blk(x)
        return x


class UNetDecoder(nn.Module):
    """DT:517-629.  `self.encoder = encoder` is kept (DT:549) so the state_dict carries the `decoder.encoder.*` aliases."""

    def __init__(self, encoder, num_classes, n_conv_per_stage, deep_supervision, nonlin_first=False, norm_op=None,
                 norm_op_kwargs=None, dropout_op=None, dropout_op_kwargs=None, nonlin=None, nonlin_kwargs=None, conv_bias=None):
        super().__init__()
        self.deep_supervision = deep_supervision
        self.encoder = encoder
        self.num_classes = num_classes
        n_stages_encoder = len(encoder.output_channels)
        if isinstance(n_conv_per_stage, int):
            n_conv_per_stage = [n_conv_per_stage] * (n_stages_encoder - 1)
        assert len(n_conv_per_stage) == n_stages_encoder - 1
        assert encoder.conv_op is nn.Conv2d, "2D only"
        conv_bias = encoder.conv_bias if conv_bias is None else conv_bias
        norm_op = encoder.norm_op if norm_op is None else norm_op
        norm_op_kwargs = encoder.norm_op_kwargs if norm_op_kwargs is None else norm_op_kwargs
        dropout_op = encoder.dropout_op if dropout_op is None else dropout_op
        dropout_op_kwargs = encoder.dropout_op_kwargs if dropout_op_kwargs is None else dropout_op_kwargs
        nonlin = encoder.nonlin if nonlin is None else nonlin
        nonlin_kwargs = encoder.nonlin_kwargs if nonlin_kwargs is None else nonlin_kwargs
        stages, transpconvs, seg_layers = [], [], []
        for s in range(1, n_stages_encoder):
            below = encoder.output_channels[-s]
            skip = encoder.output_channels[-(s + 1)]
            st = encoder.strides[-s]
            assert list(st) == [2, 2]
            transpconvs.append(nn.ConvTranspose2d(below, skip, st, st, bias=conv_bias))
            stages.append(StackedConvBlocks(n_conv_per_stage[s - 1], encoder.conv_op, 2 * skip, skip, encoder.kernel_sizes[-(s + 1)], 1,
                                            conv_bias, norm_op, norm_op_kwargs, dropout_op, dropout_op_kwargs, nonlin, nonlin_kwargs,
                                            nonlin_first))
            seg_layers.append(encoder.conv_op(skip, num_classes, 1, 1, 0, bias=True))
        self.stages = nn.ModuleList(stages)
        self.transpconvs = nn.ModuleList(transpconvs)
        self.seg_layers = nn.ModuleList(seg_layers)

    def forward(self, skips):
        """skips: NHWC tensors.  Returns fp32 NCHW logits (list, largest first, when deep supervision is on)."""
        lres = skips[-1]
        segs = []
        n = len(self.stages)
        for s in range(n):
            up = ops.conv_transpose2x2(lres, self.transpconvs[s].weight, self.transpconvs[s].bias)
            x = self.stages[s](up, skips[-(s + 2)])          # concat (DT:614) fused into the conv's two-pointer gather
            if self.deep_supervision or s == n - 1:
                sl = self.seg_layers[s if self.deep_supervision else -1]
                # the K-class head runs as a GEMM with its output columns zero-padded to a multiple of 8 (16-byte rows for the
                # vectorised dgrad/wgrad loads); the logits are the first K columns, in fp32
                K = sl.weight.shape[0]
                if ops.seg_head_ok(x, K):                   # 32 channels -> <= 4 classes: one streaming pass, NCHW fp32 logits directly
                    segs.append(ops.seg_head(x, sl.weight, sl.bias))
                    lres = x
                    continue
                Kp = (K + 7) // 8 * 8
                w8 = torch.nn.functional.pad(sl.weight.flatten(1), (0, 0, 0, Kp - K))
                b8 = torch.nn.functional.pad(sl.bias, (0, Kp - K))
                y8 = ops.conv1x1(x, w8, b8, out_dtype=torch.float32)
                segs.append(ops.nhwc_to_nchw_f32(y8[..., :K]))
            lres = x
        segs = segs[::-1]
        return segs if self.deep_supervision else segs[0]

    def compute_conv_feature_map_size(self, input_size):
        return 0


class DinoUNet(nn.Module):
    """U-Net with the DINOv3 adapter as encoder (DT:632-829)."""

    def __init__(self, network_config: dict = None, input_channels: int = None, num_classes: int = None,
                 dinov3_pretrained_path: str = "dinounet/checkpoints/dinov3_vits16_pretrain_lvd1689m-08c60483.pth",
                 dinov3_model_name: str = "dinov3_vits16", adapter_type: str = "default", n_stages: int = None,
                 features_per_stage=None, conv_op=None, kernel_sizes=None, strides=None, n_conv_per_stage=None,
                 n_conv_per_stage_decoder=None, conv_bias: bool = False, norm_op=None, norm_op_kwargs: dict = None,
                 dropout_op=None, dropout_op_kwargs: dict = None, nonlin=None, nonlin_kwargs: dict = None,
                 deep_supervision: bool = False, nonlin_first: bool = False, precision: str = None):
        super().__init__()
        if network_config is not None:                                                 # DT:664-694
            arch = network_config["architecture"]
            _res = lambda v: pydoc.locate(v) if isinstance(v, str) else v
            input_channels = input_channels or 3
            self.adapter_type = adapter_type
            num_classes = num_classes or 2
            n_stages = arch["n_stages"]
            features_per_stage = arch["features_per_stage"]
            conv_op = _res(arch["conv_op"])
            kernel_sizes = arch["kernel_sizes"]
            strides = arch["strides"]
            n_conv_per_stage = arch["n_conv_per_stage"]
            n_conv_per_stage_decoder = arch["n_conv_per_stage_decoder"]
            conv_bias = arch.get("conv_bias", False)
            norm_op = _res(arch["norm_op"])
            norm_op_kwargs = arch.get("norm_op_kwargs", {})
            dropout_op = _res(arch["dropout_op"])
            dropout_op_kwargs = arch.get("dropout_op_kwargs", {})
            nonlin = _res(arch["nonlin"])
            nonlin_kwargs = arch.get("nonlin_kwargs", {})
            deep_supervision = arch.get("deep_supervision", False)
            nonlin_first = arch.get("nonlin_first", False)
        if isinstance(n_conv_per_stage_decoder, int):
            n_conv_per_stage_decoder = [n_conv_per_stage_decoder] * (n_stages - 1)
        if n_stages != 4:                                                              # DT:703-711
            n_stages = 4
            if isinstance(features_per_stage, int):
                features_per_stage = [features_per_stage * (2 ** i) for i in range(4)]
            elif len(features_per_stage) != 4:
                base = features_per_stage[0] if features_per_stage else 32
                features_per_stage = [base * (2 ** i) for i in range(4)]
            n_conv_per_stage_decoder = (list(n_conv_per_stage_decoder) + [2, 2, 2])[:3]
        if dinov3_model_name not in DINOv3_MODEL_INFO:
            raise ValueError(f"Unknown model: {dinov3_model_name}")
        backbone = load_dinov3_model(dinov3_model_name, dinov3_pretrained_path)
        adapter = DINOv3_Adapter(backbone=backbone, interaction_indexes=DINOv3_INTERACTION_INDEXES[dinov3_model_name],
                                 pretrain_size=512, conv_inplane=64, n_points=4, deform_num_heads=16, drop_path_rate=0.3,
                                 init_values=0.0, with_cffn=True, cffn_ratio=0.25, deform_ratio=0.5, add_vit_feature=True,
                                 use_extra_extractor=True, with_cp=True)                # DT:754-769
        self.encoder = DINOv3EncoderAdapter(dinov3_adapter=adapter, target_channels=features_per_stage, conv_op=conv_op,
                                            norm_op=norm_op, norm_op_kwargs=norm_op_kwargs, dropout_op=dropout_op,
                                            dropout_op_kwargs=dropout_op_kwargs, nonlin=nonlin, nonlin_kwargs=nonlin_kwargs,
                                            conv_bias=conv_bias)
        self.decoder = UNetDecoder(self.encoder, num_classes, n_conv_per_stage_decoder, deep_supervision, nonlin_first=nonlin_first)
        self.set_precision(precision or os.environ.get("DINOUNET_PRECISION", "bf16"))

    def set_precision(self, precision: str):
        """'bf16': bf16 activations / fp32 accumulate (throughput mode).  'fp32': every kernel in its fp32 instantiation
        (parity mode: logits within 1e-3 of the reference's CPU fp32 path)."""
        if precision not in ("bf16", "fp32"):
            raise ValueError(precision)
        self.precision = precision
        dt = torch.bfloat16 if precision == "bf16" else torch.float32
        for m in self.modules():
            m._act_dtype = dt
        return self

    # nnUNetTrainer.py:210-212 may wrap the network in torch.compile: the forward is a chain of opaque custom autograd Functions over
    # the C ABI (raw pointers, ctypes), nothing dynamo can trace -- it is excluded from tracing, so the compiled wrapper runs this very
    # code (same logits, same gradients; tests/test_gpu_boundary.py::test_torch_compile_wrapper_keeps_logits)
    @torch.compiler.disable
    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("DinoUNet (dinounet_amd) runs on the MI355X through libdinounet_hip.so; move the module and "
                               "its input to the GPU (there is no CPU fallback)")
        ops.PACK.refresh()          # one launch: kernel-ready bf16 forms of every trainable weight for this step
        ops.ZEROS.new_step()        # one fill: the fp32 accumulators of this step's split-K weight gradients
        return self.decoder(self.encoder(x.float()))

    def compute_conv_feature_map_size(self, input_size):
        return 0

    @staticmethod
    def initialize(module):
        if isinstance(module, (nn.Conv2d, nn.ConvTranspose2d)):
            nn.init.kaiming_normal_(module.weight, a=1e-2)
            if module.bias is not None:
                nn.init.constant_(module.bias, 0)

    @classmethod
    def from_config(cls, network_config: dict, input_channels: int, num_classes: int,
                    dinov3_pretrained_path: str = "dinov3_vits16_pretrain_lvd1689m-08c60483.pth",
                    dinov3_model_name: str = "dinov3_vits16", **kw):
        return cls(network_config=network_config, input_channels=input_channels, num_classes=num_classes,
                   dinov3_pretrained_path=dinov3_pretrained_path, dinov3_model_name=dinov3_model_name, **kw)
