"""world_size-2 gloo tests (CPU) of the data-parallel path: bucketed gradient all-reduce (dinounet_amd.parallel) and the
DDP batch-Dice loss (all-gather forward / all-reduce backward, ddp_allgather.py:25-48)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn


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
    # DDP batch-dice: loss/grad must equal the single-process loss on the concatenated batch
    lg = torch.randn(4, 3, 8, 8, generator=torch.Generator().manual_seed(2)).requires_grad_(True)
    tg = torch.randint(0, 3, (4, 1, 8, 8), generator=torch.Generator().manual_seed(3))
    lo = lg[rank * 2:(rank + 1) * 2]
    l = dc_and_ce_loss(lo, tg[rank * 2:(rank + 1) * 2], ddp=True)
    (gl,) = torch.autograd.grad(l, lg)
    dist.all_reduce(gl)
    q.put((rank, grads, float(l), gl / world))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_and_ddp_dice_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
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
