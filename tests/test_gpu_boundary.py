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
