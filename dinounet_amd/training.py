"""Synthetic-data training-step harness replicating the reference trainer's `train_step`
(dinounet/training/nnUNetTrainer/nnUNetTrainer.py:899-929): forward -> DC+CE loss -> backward -> clip_grad_norm_(12)
-> SGD(nesterov, momentum 0.99, wd 3e-5).  The loss is the reference's formula on the fp32 logits (SURVEY.md 8f "next"
row: not yet a fused kernel, stock torch reductions)."""
import torch
import torch.distributed as dist
import torch.nn.functional as F


class _AllReduceSumGrad(torch.autograd.Function):
    """dinounet/utilities/ddp_allgather.py:25-48 followed by .sum(0): all-gather forward / all-reduce backward."""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        y = x.clone()
        dist.all_reduce(y, group=group)
        return y

    @staticmethod
    def backward(ctx, g):
        g = g.clone()
        dist.all_reduce(g, group=ctx.group)
        return g, None


def dc_and_ce_loss(logits, target, smooth=1e-5, ddp=None, group=None):
    """DC_and_CE_loss (compound_losses.py:8-56) with MemoryEfficientSoftDiceLoss(batch_dice=True, do_bg=False,
    smooth=1e-5) (dice.py:58-119, trainer config nnUNetTrainer.py:363-365).  target (B,1,H,W) integer labels."""
    K = logits.shape[1]
    lab = target[:, 0].long()
    ce = F.cross_entropy(logits, lab)
    prob = torch.softmax(logits, 1)
    with torch.no_grad():
        onehot = torch.zeros(prob.shape, device=prob.device, dtype=torch.bool).scatter_(1, lab[:, None], 1)[:, 1:]
        sum_gt = onehot.sum((2, 3))
    p = prob[:, 1:]
    inter = (p * onehot).sum((2, 3))
    sum_pred = p.sum((2, 3))
    if ddp is None:
        ddp = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    inter, sum_pred, sum_gt = inter.sum(0), sum_pred.sum(0), sum_gt.sum(0).to(p.dtype)
    if ddp:
        inter = _AllReduceSumGrad.apply(inter, group)
        sum_pred = _AllReduceSumGrad.apply(sum_pred, group)
        sum_gt = _AllReduceSumGrad.apply(sum_gt, group)
    dc = (2 * inter + smooth) / torch.clip(sum_gt + sum_pred + smooth, 1e-8)
    return ce - dc.mean()
