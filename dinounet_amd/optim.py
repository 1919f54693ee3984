"""Fused gradient clipping + Nesterov SGD: the optimiser half of the reference train step
(dinounet/training/nnUNetTrainer/nnUNetTrainer.py:486 `torch.optim.SGD(..., momentum=0.99, nesterov=True)`, :922
`clip_grad_norm_(parameters, 12)`, :924 `optimizer.step()`) as three HIP launches over all trainable tensors (csrc/optim.hip)
instead of ~20 multi-tensor launches.

`FusedClipSGD` is a `torch.optim.Optimizer`: `param_groups[0]['lr']` is read on every step (nnU-Net's PolyLRScheduler writes it
there), momentum buffers live in `state[p]['momentum_buffer']` like torch's SGD, so optimiser checkpoints interchange.  `step()`
clips to `max_norm` first (None / inf: no clipping); a preceding `torch.nn.utils.clip_grad_norm_` call, as in the reference loop, is
harmless.  Hyper-parameters travel through a pinned host buffer -> device copy, so a hipGraph-captured step follows the
learning-rate schedule: call `refresh_hyper()` before each replay (training.TrainStep does)."""
import math

import torch

from . import _lib

_CHUNK = 4096


class FusedClipSGD(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-2, momentum=0.99, weight_decay=3e-5, nesterov=True, max_norm=12.0, dampening=0.0):
        if dampening != 0.0:
            raise ValueError("FusedClipSGD implements dampening = 0 (the reference's setting)")
        if nesterov and momentum <= 0:
            raise ValueError("Nesterov momentum requires a momentum")
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay, nesterov=nesterov, dampening=0.0))
        if len(self.param_groups) != 1:
            raise ValueError("FusedClipSGD takes a single parameter group (the clip norm is global)")
        self.max_norm = max_norm
        self.total_norm = None        # device scalar of the last step (pre-clip gradient norm), like clip_grad_norm_'s return value
        self._sig = None
        self._tab_host = self._tab_host_cap = self._tab_dev = self._pre_dev = None
        self._hyper_host = self._hyper_dev = None
        self._hyper_evt = None
        self._tab_evt = [None, None]
        self._flip = 0
        self._retired = []
        self._nblocks = 0

    # ------------------------------------------------------------------------------------------------
    def refresh_hyper(self):
        """param_groups -> pinned host buffer (the device copy is part of step(), hence of a captured graph)."""
        g = self.param_groups[0]
        mn = self.max_norm
        h = self._hyper_host
        h[0], h[1], h[2] = float(g["lr"]), float(g["momentum"]), float(g["weight_decay"])
        h[3] = float("inf") if mn is None or math.isinf(mn) else float(mn)
        h[4] = 1.0 if g["nesterov"] else 0.0

    def load_state_dict(self, state_dict):
        """torch replaces state[p]['momentum_buffer'] by new tensors here: the cached device-pointer table must be rebuilt."""
        super().load_state_dict(state_dict)
        self._sig = None

    def __setstate__(self, state):
        super().__setstate__(state)
        self._sig = None

    def _tables(self, active, dev):
        # the table holds raw device pointers: its signature covers the storage of every parameter AND momentum buffer, so
        # load_state_dict / net.to() / a reallocated p.data can never leave it pointing at freed memory
        for p in active:
            st = self.state[p]
            if "momentum_buffer" not in st or st["momentum_buffer"] is None:
                st["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        sig = (tuple((id(p), p.data_ptr(), self.state[p]["momentum_buffer"].data_ptr()) for p in active), str(dev))
        if sig != self._sig:
            n = len(active)
            pre = torch.zeros(n + 1, dtype=torch.int64)
            for i, p in enumerate(active):
                pre[i + 1] = pre[i] + (p.numel() + _CHUNK - 1) // _CHUNK
            self._nblocks = int(pre[n])
            self._pre_dev = pre.to(dev)
            # two pinned tables for eager steps (the host may run a step ahead of the GPU: the copy of step k must not read what step
            # k+1 writes), one more that a captured graph keeps re-reading; tables of an earlier signature stay alive in _retired
            # because a graph captured back then still copies from their pinned addresses
            self._retired = getattr(self, "_retired", []) + [t for t in (self._tab_host_cap,) if t is not None]
            self._tab_host = [torch.zeros((n, 4), dtype=torch.int64).pin_memory() for _ in range(2)]
            self._tab_evt = [None, None]
            self._flip = 0
            self._tab_host_cap = torch.zeros((n, 4), dtype=torch.int64).pin_memory()   # the table a captured graph keeps reading
            self._tab_dev = torch.zeros((n, 4), dtype=torch.int64, device=dev)
            for i, p in enumerate(active):
                st = self.state[p]
                for t in (*self._tab_host, self._tab_host_cap):
                    t[i, 0] = p.data_ptr()
                    t[i, 2] = st["momentum_buffer"].data_ptr()
                    t[i, 3] = p.numel()
            if self._hyper_host is None or self._hyper_dev.device != dev:
                self._hyper_host = torch.zeros(8, dtype=torch.float32).pin_memory()
                self._hyper_dev = torch.zeros(8, dtype=torch.float32, device=dev)
            self._sig = sig
            # a hipGraph captured against the previous tables / momentum buffers copies from and updates memory this optimizer no longer
            # owns: holders of such a graph (training.TrainStep) compare this counter before every replay and capture again on a change
            self.generation = getattr(self, "generation", 0) + 1
        # a hipGraph capture records the host->device copy of the table; replays re-read the pinned source, so the capture gets its own
        # (its gradient addresses, from the graph's private pool, stay valid) and later eager steps cannot overwrite it
        capturing = torch.cuda.is_current_stream_capturing()
        if capturing:
            t = self._tab_host_cap
        else:
            self._flip ^= 1
            t = self._tab_host[self._flip]
            if self._tab_evt[self._flip] is not None:
                self._tab_evt[self._flip].synchronize()       # the copy that last read this pinned table (two steps ago) has finished
        t.numpy()[:, 1] = [p.grad.data_ptr() for p in active]         # one vectorised write (285 indexed tensor writes cost ~1 ms of host time)
        self._tab_dev.copy_(t, non_blocking=True)
        if not capturing:
            ev = torch.cuda.Event()
            ev.record()
            self._tab_evt[self._flip] = ev

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        active = [p for p in self.param_groups[0]["params"] if p.grad is not None]
        if not active:
            return loss
        dev = active[0].device
        if dev.type != "cuda":
            raise RuntimeError("FusedClipSGD runs on the MI355X through libdinounet_hip.so (no CPU fallback); use torch.optim.SGD on the CPU")
        for p in active:
            if p.dtype != torch.float32 or p.grad.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous() or p.device != dev:
                raise RuntimeError("FusedClipSGD needs contiguous fp32 parameters and gradients on one device")
        self._tables(active, dev)
        if not torch.cuda.is_current_stream_capturing() and self._hyper_evt is not None:
            self._hyper_evt.synchronize()                      # the previous step's copy of the pinned hyper-parameters is done
        self.refresh_hyper()
        self._hyper_dev.copy_(self._hyper_host, non_blocking=True)
        if not torch.cuda.is_current_stream_capturing():
            self._hyper_evt = torch.cuda.Event()
            self._hyper_evt.record()
        L = _lib.lib()
        n_ws = int(L.du_clip_sgd_ws_elems(self._nblocks))
        ws = torch.empty(n_ws, dtype=torch.float32, device=dev)
        _lib.check(L.du_clip_sgd(self._tab_dev.data_ptr(), self._pre_dev.data_ptr(), len(active), self._nblocks, self._hyper_dev.data_ptr(),
                                 ws.data_ptr(), n_ws, torch.cuda.current_stream().cuda_stream), "du_clip_sgd")
        self.total_norm = ws[self._nblocks]
        # the kernels wrote the parameters through raw pointers (no autograd version bump): packed / bf16 copies of them are stale
        from . import ops
        ops.PACK.invalidate()
        return loss
