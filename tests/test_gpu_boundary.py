"""The trainer-side contract of the drop-in boundary, exercised the way nnUNetTrainer uses the network:
`autocast` + `GradScaler` around the module and the loss (nnUNetTrainer.py:905-929), `SyncBatchNorm.convert_sync_batchnorm` + `DDP(...)`
(:217-218), `torch.no_grad()` validation forwards (:959).  The module is autocast-agnostic (it picks its own activation dtype) and must
simply keep working -- and keep producing the same numbers -- inside those wrappers."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _net(precision="fp32", seed=0, train=True):
    from oracle import weights
    from oracle.refshim import PLANS_2D
    from dinounet_amd.network_architecture import DinoUNet
    from dinounet_amd.dinov3.adapter import DropPath
    net = DinoUNet.from_config(PLANS_2D, 3, 2, dinov3_pretrained_path=None, dinov3_model_name="dinounet_s", precision=precision)
    ks = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    net.load_state_dict(weights.make_state_dict(ks, seed=seed), strict=True)
    net = net.cuda().train(train)
    for m in net.modules():
        if isinstance(m, DropPath):
            m.drop_prob = 0.0
    net.encoder.dinov3_adapter.backbone.rope_embed.rescale_coords = None
    return net


def _batch():
    from oracle import weights
    return weights.make_input(2, 3, 64, 64, seed=5).cuda(), weights.make_target(2, 64, 64, 2, seed=5).cuda()


def _grads(net):
    return {k: p.grad.detach().float().cpu().clone() for k, p in net.named_parameters() if p.grad is not None}


def test_autocast_and_gradscaler_train_step_like_the_trainer():
    """nnUNetTrainer.train_step: forward + loss under autocast, GradScaler.scale(l).backward(), unscale_, clip 12, step, update.
    Gradients after unscale_ equal the plain step's, the optimizer moves the trainable parameters, nothing is inf / nan."""
    from dinounet_amd.training import dc_and_ce_loss
    x, t = _batch()
    ref = _net()
    dc_and_ce_loss(ref(x), t).backward()
    g_ref = _grads(ref)

    net = _net()
    params = [p for p in net.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=1e-2, momentum=0.99, nesterov=True, weight_decay=3e-5)
    scaler = torch.amp.GradScaler("cuda")
    before = {k: p.detach().clone() for k, p in net.named_parameters() if p.requires_grad}
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", enabled=True):
        out = net(x)
        loss = dc_and_ce_loss(out, t)
    assert out.dtype == torch.float32                       # the logits contract: fp32 NCHW whatever autocast says
    scaler.scale(loss).backward()
    scaler.unscale_(opt)
    g = _grads(net)
    assert set(g) == set(g_ref)
    gmax = max(float(v.norm()) for v in g_ref.values())
    for k in g:
        assert torch.isfinite(g[k]).all(), k
        assert float((g[k] - g_ref[k]).norm()) <= 2e-3 * max(float(g_ref[k].norm()), 1e-3 * gmax), k
    torch.nn.utils.clip_grad_norm_(net.parameters(), 12)
    scaler.step(opt)
    scaler.update()
    moved = sum(int(not torch.equal(before[k], p.detach())) for k, p in net.named_parameters() if p.requires_grad and p.grad is not None)
    assert moved == len(g)
    assert all(torch.isfinite(p).all() for p in net.parameters())


def test_convert_sync_batchnorm_and_ddp_wrapper_single_rank():
    """TRN:217-218: SyncBatchNorm.convert_sync_batchnorm(network) then DDP(network, device_ids=[rank]).  The converted module keeps its
    state_dict keys and numbers; wrapped in torch DDP (gloo, one rank) a step gives the gradients of the bare module."""
    import torch.distributed as dist
    from dinounet_amd.training import dc_and_ce_loss
    x, t = _batch()
    ref = _net()
    y_ref = ref(x)
    dc_and_ce_loss(y_ref, t).backward()
    g_ref = _grads(ref)

    net = _net()
    keys = list(net.state_dict().keys())
    net = torch.nn.SyncBatchNorm.convert_sync_batchnorm(net)
    assert list(net.state_dict().keys()) == keys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        ddp = torch.nn.parallel.DistributedDataParallel(net, device_ids=[0], find_unused_parameters=True)
        y = ddp(x)
        dc_and_ce_loss(y, t).backward()
        torch.cuda.synchronize()
        assert float((y - y_ref).abs().max()) <= 1e-5 * float(y_ref.abs().max())
        g = _grads(net)
        assert set(g) == set(g_ref)
        gmax = max(float(v.norm()) for v in g_ref.values())
        for k in g:
            assert float((g[k] - g_ref[k]).norm()) <= 2e-3 * max(float(g_ref[k].norm()), 1e-3 * gmax), k
    finally:
        dist.destroy_process_group()


def test_no_grad_validation_forward():
    """TRN:959: validation_step runs the network under torch.no_grad(); same logits as the grad-enabled forward, no autograd graph."""
    x, _ = _batch()
    net = _net(train=False)
    with torch.no_grad():
        y0 = net(x)
    assert not y0.requires_grad
    y1 = net(x)
    assert torch.equal(y0, y1.detach())
    with torch.no_grad(), torch.autocast("cuda", enabled=True):
        y2 = net(x)
    assert y2.dtype == torch.float32 and torch.equal(y0, y2)


@pytest.mark.parametrize("autocast_dtype", [None, torch.float16, torch.bfloat16])
def test_reference_msdeformattn_function_unmodified_on_dropin_module(autocast_dtype):
    """the reference's OWN `MSDeformAttnFunction` (ms_deform_attn.py:28-68, vendored verbatim in
    tests/ref_vendor/ms_deform_attn_ref.py: custom_fwd(cast_inputs=fp32) forward through grid_sample, once_differentiable backward through
    `MSDA.ms_deform_attn_backward(..., im2col_step)`) runs on top of the drop-in `MultiScaleDeformableAttention` module: all three
    gradients against plain autograd through the reference's `ms_deform_attn_core_pytorch` in fp64 (what ops/test.py:101-111 checks the
    CUDA kernel against), at the production shape (Lq 5376, 32 x 32 plane, 16 heads x 32), with and without the trainer's autocast."""
    from tests.ref_vendor.ms_deform_attn_ref import MSDeformAttnFunction, ms_deform_attn_core_pytorch
    import MultiScaleDeformableAttention as MSDA
    import tests.ref_vendor.ms_deform_attn_ref as R
    assert R.MSDA is MSDA and MSDA.__file__.endswith("MultiScaleDeformableAttention.py")
    d = torch.device("cuda")
    N, S, M, D, Lq, P = 2, 1024, 16, 32, 5376, 4
    gen = torch.Generator().manual_seed(21)
    value = (torch.randn(N, S, M, D, generator=gen) * 0.5).to(d)
    loc = (torch.rand(N, Lq, M, 1, P, 2, generator=gen) * 1.1 - 0.05).to(d)
    attn = torch.softmax(torch.randn(N, Lq, M, 1, P, generator=gen), -1).to(d)
    go = torch.randn(N, Lq, M * D, generator=gen).to(d)
    shapes = torch.tensor([[32, 32]], dtype=torch.long, device=d)
    lsi = torch.zeros(1, dtype=torch.long, device=d)
    v, l, a = (t.clone().requires_grad_(True) for t in (value, loc, attn))
    if autocast_dtype is None:
        out = MSDeformAttnFunction.apply(v, shapes, lsi, l, a, 64)
    else:
        with torch.autocast("cuda", dtype=autocast_dtype):
            # inputs in the autocast dtype, as they leave the adapter's Linear layers under the trainer's autocast (TRN:914)
            out = MSDeformAttnFunction.apply(v.to(autocast_dtype), shapes, lsi, l.to(autocast_dtype), a.to(autocast_dtype), 64)
    assert out.dtype == torch.float32 and out.shape == (N, Lq, M * D)
    gv, gl, ga = torch.autograd.grad(out, (v, l, a), go)
    q = (lambda t: t.to(autocast_dtype).double()) if autocast_dtype is not None else (lambda t: t.double())
    v64, l64, a64 = (q(t).requires_grad_(True) for t in (value, loc, attn))
    ref = ms_deform_attn_core_pytorch(v64, [(32, 32)], l64, a64)
    rv, rl, ra = torch.autograd.grad(ref, (v64, l64, a64), go.double())

    def rel(x, y, mask=None):
        d_ = (x.double() - y).abs()
        if mask is not None:
            d_ = d_ * mask
        return float(d_.max() / y.abs().max())
    # The location gradient is discontinuous where a sample sits exactly on a pixel centre line (the bilinear corners switch): the
    # half / bfloat16 grids of the autocast cases put ~1 sample in 64 / 8 there, and fp32 vs fp64 round-off then picks different sides.
    # Those samples are left out of the grad_loc comparison (they are compared through grad_value / grad_attn, which are continuous).
    pix = l64.detach() * 32.0 - 0.5
    off_kink = ((pix - pix.round()).abs() > 1e-4).all(-1, keepdim=True).expand_as(pix).double()
    errs = dict(out=rel(out, ref), grad_value=rel(gv, rv), grad_loc=rel(gl, rl, off_kink), grad_attn=rel(ga, ra))
    print(f"[reference MSDeformAttnFunction on the drop-in module, autocast {autocast_dtype}] {errs}; samples off the kinks: "
          f"{float(off_kink.mean()):.4f}")
    # under autocast the gradients travel back through the .to(half) casts of the test's inputs: one rounding to the autocast dtype
    tol = {None: 1e-5, torch.float16: 2e-3, torch.bfloat16: 1.6e-2}[autocast_dtype]
    assert errs["out"] < 1e-5 and errs["grad_value"] < tol and errs["grad_attn"] < tol and errs["grad_loc"] < max(tol, 2e-4), errs
    # once_differentiable (MSA:48): a second derivative through the backward must raise, as with the compiled extension
    v2 = value.clone().requires_grad_(True)
    o2 = MSDeformAttnFunction.apply(v2, shapes, lsi, loc, attn, 64)
    g2, = torch.autograd.grad(o2, v2, go, create_graph=True)
    with pytest.raises(RuntimeError):
        g2.sum().backward()
    # the extension's own argument check (ms_deform_attn_cuda.cu:57): batch not divisible by min(batch, im2col_step)
    with pytest.raises(RuntimeError):
        MSDA.ms_deform_attn_backward(torch.cat([value, value[:1]]).contiguous(), shapes, lsi, torch.cat([loc, loc[:1]]).contiguous(),
                                     torch.cat([attn, attn[:1]]).contiguous(), torch.cat([go, go[:1]]).contiguous(), 2)


def test_torch_compile_wrapper_keeps_logits():
    """nnUNetTrainer.py:210-212: `self.network = torch.compile(self.network)` when nnUNet_compile is set.  The module is a graph of opaque
    custom autograd Functions over the C ABI: dynamo must break the graph around them (or fall back to eager) and return the same logits
    and gradients as the bare module -- it does not have to speed anything up."""
    import torch._dynamo
    from dinounet_amd.training import dc_and_ce_loss
    x, t = _batch()
    ref = _net()
    y_ref = ref(x)
    dc_and_ce_loss(y_ref, t).backward()
    g_ref = _grads(ref)
    torch._dynamo.reset()
    net = _net()
    cnet = torch.compile(net)
    y = cnet(x)
    assert y.dtype == torch.float32 and y.shape == y_ref.shape
    assert float((y - y_ref).abs().max()) <= 1e-5 * float(y_ref.abs().max())
    dc_and_ce_loss(y, t).backward()
    g = _grads(net)
    assert set(g) == set(g_ref)
    gmax = max(float(v.norm()) for v in g_ref.values())
    for k in g:
        assert float((g[k] - g_ref[k]).norm()) <= 2e-3 * max(float(g_ref[k].norm()), 1e-3 * gmax), k
    net.eval()
    with torch.no_grad():
        y_eval = torch.compile(net)(x)
    assert torch.isfinite(y_eval).all()
    torch._dynamo.reset()


def test_reference_ops_test_script_as_written_on_dropin_module():
    """The reference extension's OWN acceptance script (ops/test.py, vendored whole and unmodified as tests/ref_vendor/ops_test_ref.py
    together with the autograd Function it imports, ops/functions/ms_deform_attn_func.py: forward AND backward through the compiled
    module) executed as a script on top of the drop-in `MultiScaleDeformableAttention`: check_forward_equal_with_pytorch_double
    (torch.allclose at default tolerances, .double() tensors: ops/test.py:40-58), check_forward_equal_with_pytorch_float (:61-83) and
    torch.autograd.gradcheck in double for D in {30, 32, 64, 71, 1025, 2048, 3096} (:86-121).  Needs the fp64 dispatch of the reference
    (AT_DISPATCH_FLOATING_TYPES, ms_deform_attn_cuda.cu:69,139) = du_msda_forward_f64 / du_msda_backward_f64."""
    import os
    import subprocess
    import sys
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_vendor")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "ops_test_ref.py"], cwd=here, env=env, capture_output=True, text=True, timeout=1500)
    print(r.stdout[-3000:], r.stderr[-3000:])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("* ")]
    assert len(lines) == 2 + 7, lines
    assert all(ln.startswith("* True") for ln in lines), lines


@pytest.mark.parametrize("variant", ["dropout_p0", "nonlin_first", "leaky_slope", "gelu", "in_no_affine", "groupnorm", "no_norm"])
def test_decoder_block_variants_the_constructor_accepts(variant):
    """ConvDropoutNormReLU / StackedConvBlocks with the arguments the 2D plans never use but the reference's constructor takes
    (dinounet_training.py:581-592: dropout_op, nonlin_first, other activations / norms): composed from the same kernels plus the torch
    module for the piece that has no kernel, against the plain torch composition in the same order, forward and gradients."""
    from torch import nn
    import torch.nn.functional as F
    from dinounet_amd.network_architecture.dinounet import ConvDropoutNormReLU
    kw = dict(conv_bias=True, norm_op=nn.InstanceNorm2d, norm_op_kwargs={"eps": 1e-5, "affine": True}, nonlin=nn.LeakyReLU,
              nonlin_kwargs={"inplace": True})
    if variant == "dropout_p0":
        kw.update(dropout_op=nn.Dropout2d, dropout_op_kwargs={"p": 0.0})
    elif variant == "nonlin_first":
        kw.update(nonlin_first=True)
    elif variant == "leaky_slope":
        kw.update(nonlin_kwargs={"negative_slope": 0.2})
    elif variant == "gelu":
        kw.update(nonlin=nn.GELU, nonlin_kwargs={})
    elif variant == "in_no_affine":
        kw.update(norm_op_kwargs={"eps": 1e-5, "affine": False})
    elif variant == "groupnorm":
        kw.update(norm_op=lambda c, **k: nn.GroupNorm(4, c), norm_op_kwargs={})
    elif variant == "no_norm":
        kw.update(norm_op=None, norm_op_kwargs=None)
    torch.manual_seed(3)
    blk = ConvDropoutNormReLU(nn.Conv2d, 16, 32, 3, 1, **kw).cuda().train()
    x = torch.randn(2, 24, 40, 16, device="cuda", requires_grad=True)        # NHWC, fp32 mode
    y = blk(x)
    go = torch.randn_like(y)
    y.backward(go)
    got = [y.detach(), x.grad.detach()] + [p.grad.detach() for p in blk.parameters()]
    # reference: the registered modules in their registered order on the NCHW view
    for p in blk.parameters():
        p.grad = None
    xr = x.detach().clone().requires_grad_(True)
    yr = blk.all_modules(xr.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    yr.backward(go)
    want = [yr.detach(), xr.grad.detach()] + [p.grad.detach() for p in blk.parameters()]
    for a, b in zip(got, want):
        assert a.shape == b.shape
        # (+ an absolute floor: a conv bias in front of a norm has a mathematically zero gradient, ~1e-5 of rounding noise on both sides)
        assert (a - b).norm() <= 2e-4 * b.norm() + 1e-3, (variant, float((a - b).norm()), float(b.norm()))
