"""Helpers kept for name compatibility with the reference's dinounet/network_architecture/utils.py:5-29."""
import torch


def softmax_helper(x):
    return torch.softmax(x, 1)


def to_cuda(data, non_blocking=True, gpu_id=0):
    if isinstance(data, list):
        return [i.cuda(gpu_id, non_blocking=non_blocking) for i in data]
    return data.cuda(gpu_id, non_blocking=non_blocking)
