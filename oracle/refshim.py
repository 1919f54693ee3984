"""TEST INFRASTRUCTURE ONLY -- import shim that builds the *reference's own* DinoUNet on CPU.

Only usable inside the build container (needs /root/reference, which does not exist on the
GPU box).  Used by oracle/make_golden.py to pin the oracle restatement (oracle/dinounet_oracle.py)
and to generate the committed fixtures under tests/golden/.  Never imported by the product
package `dinounet_amd`.

Shims:
  1. `dinounet/__init__.py:1` imports `dinounet.api` (needs batchgenerators, SimpleITK...) ->
     pre-register an empty package object whose __path__ points at the reference tree.
  2. `ms_deform_attn.py:18` hard-imports the CUDA extension `MultiScaleDeformableAttention` ->
     stub module; backward = autograd through `ms_deform_attn_core_pytorch` (ms_deform_attn.py:71-92),
     exactly what ops/test.py:101-111 gradchecks the CUDA kernel against.
  3. `dynamic_network_architectures` (requirements.txt:3, absent) -> restatement of the three
     symbols dinounet_training.py:13-20 uses ("parity unpinned" at that boundary).
  4. `LinearKMaskedBias.bias_mask` is NaN-filled at construction (layers/attention.py:36) and only
     set by a checkpoint; we set it to 1|0|1 (K third zero) like the released checkpoints do.
"""
import os
import sys
import types

import torch
from torch import nn

REF_ROOT = os.environ.get("DINOUNET_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "dinounet", "dinov3"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _ConvDropoutNormReLU(nn.Module):
    """Restatement of dynamic_network_architectures.building_blocks.simple_conv_blocks.ConvDropoutNormReLU
    (0.4.x): conv(k, stride, pad=(k-1)//2, bias) -> [dropout] -> norm -> nonlin, sub-modules named
    conv / norm / nonlin plus the nn.Sequential `all_modules` that aliases them."""

    def __init__(self, conv_op, input_channels, output_channels, kernel_size, stride, conv_bias=False,
                 norm_op=None, norm_op_kwargs=None, dropout_op=None, dropout_op_kwargs=None,
                 nonlin=None, nonlin_kwargs=None, nonlin_first=False):
        super().__init__()
        if not isinstance(kernel_size, (tuple, list)):
            kernel_size = [kernel_size] * 2
        if not isinstance(stride, (tuple, list)):
            stride = [stride] * 2
        ops = []
        self.conv = conv_op(input_channels, output_channels, kernel_size, stride,
                            padding=[(i - 1) // 2 for i in kernel_size], dilation=1, bias=conv_bias)
        ops.append(self.conv)
        if dropout_op is not None:
            self.dropout = dropout_op(**(dropout_op_kwargs or {}))
            ops.append(self.dropout)
        if norm_op is not None:
            self.norm = norm_op(output_channels, **(norm_op_kwargs or {}))
            ops.append(self.norm)
        if nonlin is not None:
            self.nonlin = nonlin(**(nonlin_kwargs or {}))
            ops.append(self.nonlin)
        if nonlin_first and (norm_op is not None and nonlin is not None):
            ops[-1], ops[-2] = ops[-2], ops[-1]
        self.all_modules = nn.Sequential(*ops)

    def forward(self, x):
        return self.all_modules(x)


class _StackedConvBlocks(nn.Module):
    def __init__(self, num_convs, conv_op, input_channels, output_channels, kernel_size, initial_stride,
                 conv_bias=False, norm_op=None, norm_op_kwargs=None, dropout_op=None, dropout_op_kwargs=None,
                 nonlin=None, nonlin_kwargs=None, nonlin_first=False):
        super().__init__()
        if not isinstance(output_channels, (tuple, list)):
            output_channels = [output_channels] * num_convs
        self.convs = nn.Sequential(
            _ConvDropoutNormReLU(conv_op, input_channels, output_channels[0], kernel_size, initial_stride, conv_bias,
                                 norm_op, norm_op_kwargs, dropout_op, dropout_op_kwargs, nonlin, nonlin_kwargs,
                                 nonlin_first),
            *[_ConvDropoutNormReLU(conv_op, output_channels[i - 1], output_channels[i], kernel_size, 1, conv_bias,
                                   norm_op, norm_op_kwargs, dropout_op, dropout_op_kwargs, nonlin, nonlin_kwargs,
                                   nonlin_first) for i in range(1, num_convs)])

    def forward(self, x):
        return self.convs(x)


def _msda_backward(value, shapes, level_start_index, loc, attn, grad_output, im2col_step):
    from dinounet.dinov3.eval.segmentation.models.utils.ms_deform_attn import ms_deform_attn_core_pytorch
    with torch.enable_grad():
        v = value.detach().requires_grad_(True)
        l = loc.detach().requires_grad_(True)
        a = attn.detach().requires_grad_(True)
        out = ms_deform_attn_core_pytorch(v, shapes, l, a)
        gv, gl, ga = torch.autograd.grad(out, (v, l, a), grad_output)
    return gv, gl, ga


_installed = False


def install():
    """Register the shims; idempotent."""
    global _installed
    if _installed:
        return
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REF_ROOT} (only present in the build container)")
    sys.dont_write_bytecode = True  # the reference tree is read-only
    pkg = types.ModuleType("dinounet")
    pkg.__path__ = [os.path.join(REF_ROOT, "dinounet")]
    sys.modules["dinounet"] = pkg
    _stub("dinounet.api", plan_and_preprocess=None, training=None, evaluate=None)
    for n in ("dinounet.training", "dinounet.training.nnUNetTrainer"):
        m = types.ModuleType(n)
        m.__path__ = []
        sys.modules[n] = m

    class nnUNetTrainerNoDeepSupervision:  # placeholder base (real one needs batchgenerators)
        pass

    _stub("dinounet.training.nnUNetTrainer.nnUNetTrainerNoDeepSupervision",
          nnUNetTrainerNoDeepSupervision=nnUNetTrainerNoDeepSupervision)
    _stub("MultiScaleDeformableAttention", ms_deform_attn_backward=_msda_backward, ms_deform_attn_forward=None)
    dna = types.ModuleType("dynamic_network_architectures"); dna.__path__ = []
    sys.modules["dynamic_network_architectures"] = dna
    bb = types.ModuleType("dynamic_network_architectures.building_blocks"); bb.__path__ = []
    sys.modules["dynamic_network_architectures.building_blocks"] = bb
    ini = types.ModuleType("dynamic_network_architectures.initialization"); ini.__path__ = []
    sys.modules["dynamic_network_architectures.initialization"] = ini

    def convert_conv_op_to_dim(conv_op):
        return {nn.Conv1d: 1, nn.Conv2d: 2, nn.Conv3d: 3}[conv_op]

    def get_matching_convtransp(conv_op=None, dimension=None):
        return {nn.Conv1d: nn.ConvTranspose1d, nn.Conv2d: nn.ConvTranspose2d, nn.Conv3d: nn.ConvTranspose3d}[conv_op]

    _stub("dynamic_network_architectures.building_blocks.helper",
          convert_conv_op_to_dim=convert_conv_op_to_dim, get_matching_convtransp=get_matching_convtransp)
    _stub("dynamic_network_architectures.building_blocks.plain_conv_encoder", PlainConvEncoder=nn.Module)
    _stub("dynamic_network_architectures.building_blocks.simple_conv_blocks",
          StackedConvBlocks=_StackedConvBlocks, ConvDropoutNormReLU=_ConvDropoutNormReLU)

    class InitWeights_He:
        def __init__(self, neg_slope=1e-2):
            self.neg_slope = neg_slope

        def __call__(self, module):
            if isinstance(module, (nn.Conv2d, nn.ConvTranspose2d)):
                nn.init.kaiming_normal_(module.weight, a=self.neg_slope)
                if module.bias is not None:
                    nn.init.constant_(module.bias, 0)

    _stub("dynamic_network_architectures.initialization.weight_init", InitWeights_He=InitWeights_He)
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    _installed = True


from dinounet_amd.plans import PLANS_2D  # noqa: E402,F401  (the 2D plans dict lives with the product: bench.py / tools must not import oracle/)


def build_reference_dinounet(model_name="dinounet_s", num_classes=2, seed=0, deep_supervision=False, vit_kwargs=None):
    """Instantiate the reference DinoUNet (dinounet_training.py:632) with seeded random weights."""
    install()
    import io, contextlib
    import dinounet_training as DT

    def _load(model_name_, pretrained_path=None):  # dinounet_training.py:51 without the download
        m = DT.DINOv3_MODEL_FACTORIES[model_name_](pretrained=False, **(vit_kwargs or {}))
        m.init_weights()
        for blk in m.blocks:
            qkv = blk.attn.qkv
            if hasattr(qkv, "bias_mask"):
                o = qkv.out_features // 3
                qkv.bias_mask.fill_(1.0)
                qkv.bias_mask[o:2 * o] = 0.0
        return m

    DT.load_dinov3_model = _load
    torch.manual_seed(seed)
    cfg = {"architecture": dict(PLANS_2D["architecture"], deep_supervision=deep_supervision),
           "data_config": PLANS_2D["data_config"]}
    with contextlib.redirect_stdout(io.StringIO()):
        net = DT.DinoUNet.from_config(cfg, 3, num_classes, dinov3_pretrained_path=None, dinov3_model_name=model_name)
    return net
