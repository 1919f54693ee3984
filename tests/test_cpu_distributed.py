"""world_size-2 gloo tests (CPU) of the data-parallel path: bucketed gradient all-reduce (dinounet_amd.parallel) and the
DDP batch-Dice loss (all-gather forward / all-reduce backward, ddp_allgather.py:25-48)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn


def _np(x):
    """queue payloads travel as numpy arrays, pickled BY VALUE: a torch tensor is handed over as a shared-memory handle that dies with its
    producer, and a worker that exits before the parent has opened the handle makes q.get() raise FileNotFoundError (seen once in round 5)"""
    if isinstance(x, torch.Tensor):
        return x.detach().numpy().copy()
    if isinstance(x, dict):
        return {k: _np(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(_np(v) for v in x)
    return x


def _t(x):
    import numpy as np
    if isinstance(x, np.ndarray):
        return torch.from_numpy(x)
    if isinstance(x, dict):
        return {k: _t(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(_t(v) for v in x)
    return x


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class Toy(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(6, 5)
        self.frozen = nn.Linear(5, 5)
        self.frozen.requires_grad_(False)
        self.b = nn.Linear(5, 3)
        self.never_used = nn.Linear(3, 3)       # like decoder.seg_layers.{0,1}: never receives a gradient

    def forward(self, x):
        return self.b(torch.tanh(self.frozen(torch.tanh(self.a(x)))))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dinounet_amd.parallel import GradAllReducer
    from dinounet_amd.training import dc_and_ce_loss
    torch.manual_seed(0)
    net = Toy()
    red = GradAllReducer(net, world, bucket_elems=16, skip=("never_used",))   # tiny buckets -> several collectives
    assert len(red.buckets) >= 2
    g = torch.Generator().manual_seed(1)
    X = torch.randn(8, 6, generator=g)
    T = torch.randn(8, 3, generator=g)
    xs, ts = X[rank * 4:(rank + 1) * 4], T[rank * 4:(rank + 1) * 4]
    for _ in range(2):                                                       # two steps: hooks re-arm
        for p in net.parameters():
            p.grad = None
        ((net(xs) - ts) ** 2).mean().backward()
        red.finish()
    grads = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    # the collective leg of a step captured in segments (training._CaptureSegments): hooks only gather, every bucket is all-reduced
    # after backward by reduce_deferred(), finish() divides and installs -- same averaged gradients, and the hooks re-arm
    for _ in range(2):
        for p in net.parameters():
            p.grad = None
        red.defer = True
        ((net(xs) - ts) ** 2).mean().backward()
        assert all(w is None for w in red.works)
        red.fill_missing()
        red.reduce_deferred()
        red.finish()
        red.defer = False
        for n, p in net.named_parameters():
            if p.grad is not None:
                assert torch.allclose(p.grad, grads[n], atol=1e-7), n
    # DDP batch-dice: loss/grad must equal the single-process loss on the concatenated batch
    lg = torch.randn(4, 3, 8, 8, generator=torch.Generator().manual_seed(2)).requires_grad_(True)
    tg = torch.randint(0, 3, (4, 1, 8, 8), generator=torch.Generator().manual_seed(3))
    lo = lg[rank * 2:(rank + 1) * 2]
    l = dc_and_ce_loss(lo, tg[rank * 2:(rank + 1) * 2], ddp=True)
    (gl,) = torch.autograd.grad(l, lg)
    dist.all_reduce(gl)
    q.put(_np((rank, grads, float(l.detach()), gl / world)))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_and_ddp_dice_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted([_t(q.get(timeout=120)) for _ in range(world)], key=lambda t: t[0])
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    # single-process reference on the full batch
    torch.manual_seed(0)
    net = Toy()
    g = torch.Generator().manual_seed(1)
    X = torch.randn(8, 6, generator=g); T = torch.randn(8, 3, generator=g)
    ((net(X) - T) ** 2).mean().backward()
    for rank, grads, _, _ in res:
        assert "never_used.weight" not in grads and "frozen.weight" not in grads
        for n, p in net.named_parameters():
            if p.grad is not None:
                assert torch.allclose(grads[n], p.grad, atol=1e-6), n
    # dice: CE averages per rank (mean of rank means == global mean for equal splits); dice uses global sums
    from dinounet_amd.training import dc_and_ce_loss
    lg = torch.randn(4, 3, 8, 8, generator=torch.Generator().manual_seed(2)).requires_grad_(True)
    tg = torch.randint(0, 3, (4, 1, 8, 8), generator=torch.Generator().manual_seed(3))
    l = dc_and_ce_loss(lg, tg, ddp=False)
    (gref,) = torch.autograd.grad(l, lg)
    mean_loss = sum(r[2] for r in res) / world
    assert abs(mean_loss - float(l)) < 1e-5
    assert torch.allclose(res[0][3], gref, atol=1e-6)


# ---- the CPU stand-ins of the ops a multi-rank adapter / a 7B-style train-mode backbone reach (tests/_cpu_op_shim.py): the SyncBatchNorm
#      contract over two gloo ranks, the SwiGLU gate from the interleaved projection, the DropPath sample gather / scatter ----
def _syncbn_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _cpu_op_shim as shim
    from dinounet_amd import ops
    from dinounet_amd._lib import ACT_RELU
    g = torch.Generator().manual_seed(5)
    X = [torch.randn(4, 6, 5, 8, generator=g), torch.randn(4, 3, 3, 16, generator=g)]
    DY = [torch.randn(4, 6, 5, 8, generator=g), torch.randn(4, 3, 3, 16, generator=g)]
    torch.manual_seed(7)
    bns = [nn.BatchNorm2d(8), nn.BatchNorm2d(16)]
    for bn in bns:
        nn.init.normal_(bn.weight, 1.0, 0.2); nn.init.normal_(bn.bias, 0.0, 0.2)
    xs = [x[rank * 2:(rank + 1) * 2].clone().requires_grad_(True) for x in X]
    with shim.patched_ops():
        ys = ops.sync_bn_multi(xs, bns, ACT_RELU, None)
    torch.autograd.backward(ys, [d[rank * 2:(rank + 1) * 2] for d in DY])
    q.put(_np((rank, [y.detach() for y in ys], [x.grad for x in xs], [bn.weight.grad for bn in bns], [bn.bias.grad for bn in bns],
               [bn.running_mean.clone() for bn in bns], [bn.running_var.clone() for bn in bns])))
    dist.barrier()
    dist.destroy_process_group()


def test_syncbn_stand_in_matches_full_batch_batchnorm_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_syncbn_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted([_t(q.get(timeout=120)) for _ in range(world)], key=lambda t: t[0])
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    g = torch.Generator().manual_seed(5)
    X = [torch.randn(4, 6, 5, 8, generator=g), torch.randn(4, 3, 3, 16, generator=g)]
    DY = [torch.randn(4, 6, 5, 8, generator=g), torch.randn(4, 3, 3, 16, generator=g)]
    torch.manual_seed(7)
    bns = [nn.BatchNorm2d(8), nn.BatchNorm2d(16)]
    for bn in bns:
        nn.init.normal_(bn.weight, 1.0, 0.2); nn.init.normal_(bn.bias, 0.0, 0.2)
    for j in range(2):
        x = X[j].clone().requires_grad_(True)
        y = torch.relu(bns[j](x.permute(0, 3, 1, 2))).permute(0, 2, 3, 1)          # full batch, NCHW module on the NHWC tensor
        y.backward(DY[j])
        yy = torch.cat([res[0][1][j], res[1][1][j]], 0)
        dx = torch.cat([res[0][2][j], res[1][2][j]], 0)
        assert torch.allclose(yy, y.detach(), atol=1e-5)
        assert torch.allclose(dx, x.grad, atol=1e-5)
        assert torch.allclose(res[0][3][j] + res[1][3][j], bns[j].weight.grad, atol=1e-4)    # local sums add up to the full-batch gradient
        assert torch.allclose(res[0][4][j] + res[1][4][j], bns[j].bias.grad, atol=1e-4)
        for r in range(2):
            assert torch.allclose(res[r][5][j], bns[j].running_mean, atol=1e-6)
            assert torch.allclose(res[r][6][j], bns[j].running_var, atol=1e-6)


def test_swiglu_and_sample_copy_stand_ins():
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _cpu_op_shim as shim
    from dinounet_amd import ops
    g = torch.Generator().manual_seed(3)
    x, w1, w2 = torch.randn(7, 12, generator=g), torch.randn(10, 12, generator=g), torch.randn(10, 12, generator=g)
    b1, b2 = torch.randn(10, generator=g), torch.randn(10, generator=g)
    w12 = torch.stack([w1, w2], 1).reshape(20, 12)
    b12 = torch.stack([b1, b2], 1).reshape(20)
    with shim.patched_ops():
        h = ops.mm_swiglu(x, w12, b12)
        t = torch.arange(24.0).view(4, 6)
        idx = torch.tensor([2, 0])
        sub = ops.sample_gather(t, idx)
        ops.sample_scatter_(t, sub * 10, idx)
    assert torch.allclose(h, torch.nn.functional.silu(x @ w1.t() + b1) * (x @ w2.t() + b2), atol=1e-5)
    assert torch.equal(sub, torch.arange(24.0).view(4, 6)[[2, 0]])
    assert torch.equal(t[2], torch.arange(12.0, 18.0) * 10) and torch.equal(t[1], torch.arange(6.0, 12.0))


# ---- two ranks through the whole product wiring: DinoUNet (dinounet_s) in train mode, one slice per rank, SyncBatchNorm statistics of the
#      SPM stem (one collective per norm) and of the four output norms (ONE packed collective, ops.sync_bn_multi) over gloo.  Every rank's
#      logits must equal the reference's full-batch logits of its slice (tests/golden/dinounet_s_64_train.npz: torch BatchNorm over both
#      slices), the averaged gradients the full-batch gradients, the running statistics the full-batch ones.
_WATCH = ("spm.stem.0.weight", "spm.stem.1.weight", "spm.stem.1.bias", "spm.conv4.0.weight", "dinov3_adapter.norm1.weight", "dinov3_adapter.norm4.bias",
          "dinov3_adapter.up.weight", "decoder.stages.0.convs.0.conv.weight")


def _adapter_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), DINOUNET_ALLOW_RANDOM_BACKBONE="1")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    import json
    import _cpu_op_shim as shim
    from oracle import weights
    from oracle.refshim import PLANS_2D
    from dinounet_amd.network_architecture import DinoUNet
    from dinounet_amd.dinov3.adapter import DropPath
    torch.set_num_threads(2)
    keys = json.load(open(os.path.join(here, "golden", "state_dict_dinounet_s.json")))["keys"]
    net = DinoUNet.from_config(PLANS_2D, 3, 2, dinov3_pretrained_path=None, dinov3_model_name="dinounet_s", precision="fp32")
    net.load_state_dict(weights.make_state_dict([(k, tuple(s)) for k, s, _ in keys], seed=0), strict=True)
    net.train()
    for m in net.modules():
        if isinstance(m, DropPath):
            m.drop_prob = 0.0
    net.encoder.dinov3_adapter.backbone.rope_embed.rescale_coords = None
    x = weights.make_input(2, 3, 64, 64, seed=1)
    if world > 1:
        x = x[rank:rank + 1]
    with shim.patched_ops():
        y = net.decoder(net.encoder(x))
        (y * y).mean().backward()
    named = dict(net.named_parameters())
    # numpy arrays travel through the queue by value (a tensor is handed over as a shared-memory handle its producer must outlive)
    grads = {k: named[k].grad.numpy().copy() for k in named if k.endswith(_WATCH) and named[k].grad is not None}
    bufs = {k: v.numpy().copy() for k, v in net.named_buffers()
            if ("running_mean" in k or "running_var" in k) and ("spm.stem.1" in k or "norm2" in k)}
    q.put((rank, y.detach().numpy().copy(), grads, bufs))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_dinounet_two_ranks_syncbn_through_the_adapter_matches_full_batch():
    import numpy as np
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_adapter_worker, args=(r, 2, port, q)) for r in range(2)]
    ref = ctx.Process(target=_adapter_worker, args=(0, 1, _free_port(), q))          # the full batch in ONE process, same code path
    [p.start() for p in procs + [ref]]
    got = [q.get(timeout=900) for _ in range(3)]
    [p.join(timeout=120) for p in procs + [ref]]
    assert all(p.exitcode == 0 for p in procs + [ref])
    got = [(r, torch.from_numpy(y), {k: torch.from_numpy(v) for k, v in gr.items()}, {k: torch.from_numpy(v) for k, v in bf.items()})
           for r, y, gr, bf in got]
    full = [g for g in got if g[1].shape[0] == 2][0]
    ranks = sorted([g for g in got if g[1].shape[0] == 1], key=lambda t: t[0])
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dinounet_s_64_train.npz"))
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    assert rel(full[1], torch.from_numpy(gold["logits"])) < 2e-5                     # the single-process run is the reference's
    for r in range(2):
        assert rel(ranks[r][1][0], full[1][r]) < 2e-5, r                            # each rank's slice, with the OTHER rank's statistics
    assert len(full[2]) >= 6
    # Gradient tolerance: this network amplifies input rounding ~200x into its gradients (3e-6 relative noise on the decoder's inputs moves
    # the decoder's weight gradients by 6e-4; the two setups' features differ by 3e-6: batch statistics summed in another order), so the
    # averaged gradients are held to 1e-2 of the tensor's maximum and 2e-3 in norm -- a wiring error (a missing cross-rank term, the local
    # instead of the global count) shows up at 0.1-1.
    bad = {}
    for k, g in full[2].items():
        avg = (ranks[0][2][k] + ranks[1][2][k]) / 2                                 # DDP's average of the per-rank mean losses
        if rel(avg, g) >= 1e-2 or abs(float(avg.norm()) / float(g.norm()) - 1.0) >= 2e-3:
            bad[k] = (rel(avg, g), float(g.norm()), float(avg.norm()))
    assert not bad, bad
    for k, v in full[3].items():
        for r in range(2):
            assert torch.allclose(ranks[r][3][k], v, rtol=1e-4, atol=1e-6), (k, r)


def _mode_worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from dinounet_amd.training import TrainStep
        ts = TrainStep(nn.Linear(2, 2), None, [], (1, 3, 8, 8), (1, 1, 8, 8), torch.device("cpu"), reducer=object(), graph=False)
        # rank 1's capture "fails" in the first round, both succeed in the second, both fail in the third
        res = [ts._all_ranks(rank == 0), ts._all_ranks(True), ts._all_ranks(False)]
        q.put((rank, res))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "ERR " + traceback.format_exc()))


def test_capture_mode_is_a_collective_decision_world2():
    """a rank whose whole-step capture failed must not fall back alone -- segmented replays issue the bucket all-reduces
    in bucket-index order, whole-step / eager steps in hook-ready order, so mixed modes hang or sum the wrong buckets.  TrainStep._all_ranks
    is MIN over the ranks of 'my capture worked': a rank that succeeded where another failed gets False too."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_mode_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in ps:
        p.join(timeout=60)
    for r in range(2):
        assert got[r] == [False, True, False], got
