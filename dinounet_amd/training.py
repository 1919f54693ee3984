"""Synthetic-data training-step harness replicating the reference trainer's `train_step`
(dinounet/training/nnUNetTrainer/nnUNetTrainer.py:899-929): forward -> DC+CE loss -> backward -> clip_grad_norm_(12)
-> SGD(nesterov, momentum 0.99, wd 3e-5).  On the GPU the loss is the fused Dice+CE kernel pair (csrc/loss.hip) and clip + SGD the
fused three-launch optimiser (csrc/optim.hip); the torch formula below serves the CPU / many-class path of the gloo tests."""
import os
import threading

import torch
import torch.distributed as dist
import torch.nn.functional as F

from .optim import FusedClipSGD

# test configuration (ops.sync_active): the SyncBatchNorm / batch-Dice collectives issued on a one-rank group as well
_FORCE_SMALL = os.environ.get("DINOUNET_FORCE_SMALL_COLLECTIVES") == "1"


def _ddp_default(group):
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or _FORCE_SMALL)


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
    smooth=1e-5) (dice.py:58-119, trainer config nnUNetTrainer.py:363-365).  target (B,1,H,W) integer labels.
    On the GPU (2..8 classes) this is the fused HIP loss (ops.dice_ce_loss); the torch formula below is the CPU / many-class path
    used by the gloo tests."""
    K = logits.shape[1]
    if logits.is_cuda and 2 <= K <= 8:
        from . import ops
        if ddp is None:
            ddp = _ddp_default(group)
        return ops.dice_ce_loss(logits, target, smooth, (group if group is not None else dist.group.WORLD) if ddp else None)
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
        ddp = _ddp_default(group)
    inter, sum_pred, sum_gt = inter.sum(0), sum_pred.sum(0), sum_gt.sum(0).to(p.dtype)
    if ddp:
        inter = _AllReduceSumGrad.apply(inter, group)
        sum_pred = _AllReduceSumGrad.apply(sum_pred, group)
        sum_gt = _AllReduceSumGrad.apply(sum_gt, group)
    dc = (2 * inter + smooth) / torch.clip(sum_gt + sum_pred + smooth, 1e-8)
    return ce - dc.mean()


class _CaptureSegments:
    """A train step recorded as SEVERAL hipGraphs with host-issued work between them (DINOUNET_COMM_OUTSIDE_GRAPH=1, or the fallback
    when the capture of a whole multi-rank step fails): graph | batch-Dice all-reduce | graph | gradient all-reduces | graph.  No RCCL
    call is recorded into a graph; everything else keeps the replay's launch path.  The graphs share one memory pool and are replayed
    in capture order, so a tensor made in one segment is alive in the next.  A cut ends the running capture, which HIP only allows from
    the thread that began it: cuts happen at main-thread points of the step (the loss forward, after backward), never inside autograd's
    device thread.  Consequence: a collective issued by a BACKWARD node (SyncBatchNorm's statistics all-reduce, i.e. the full model at
    N > 1) asks for a cut from that thread, `cut` raises, and TrainStep falls back to eager steps for that configuration."""

    def __init__(self, mode):
        self.mode = mode
        self.graphs, self.between, self.what = [], [], []
        self.pool = torch.cuda.graph_pool_handle()
        self.cur = None
        self.thread = threading.get_ident()

    def begin(self):
        self.cur = torch.cuda.CUDAGraph()
        self.cur.capture_begin(pool=self.pool, capture_error_mode=self.mode)

    def cut(self, eager_fn, what="cut"):
        if threading.get_ident() != self.thread and self.mode != "relaxed":
            # (hipStreamEndCapture from another thread than the one that began the capture is an error in the global / thread-local
            #  modes; "relaxed" lifts that, which is what a collective issued by a backward node -- autograd's device thread -- needs)
            raise RuntimeError(f"segmented capture: '{what}' asks for a cut from a thread that did not begin the capture")
        self.cur.capture_end()
        self.graphs.append(self.cur)
        self.between.append(eager_fn)
        self.what.append(what)
        self.begin()

    def end(self):
        self.cur.capture_end()
        self.graphs.append(self.cur)
        self.between.append(None)
        self.cur = None

    def abort(self):
        if self.cur is not None:
            try:
                self.cur.capture_end()
            except Exception:  # noqa: BLE001
                pass
            self.cur = None

    def replay(self):
        for g, fn in zip(self.graphs, self.between):
            g.replay()
            if fn is not None:
                fn()

    def reset(self):
        for g in self.graphs:
            g.reset()
        self.graphs, self.between, self.what = [], [], []


class TrainStep:
    """One optimiser step of the reference trainer (nnUNetTrainer.py:899-929) on static input buffers:
    zero_grad -> forward -> DC+CE -> backward [-> bucketed RCCL all-reduce] -> clip_grad_norm_(12) -> SGD step.

    With `graph=True` the whole step (a few thousand launches, most of them small) is captured once into a hipGraph
    after `warmup` eager iterations and replayed afterwards, which removes the host launch path from the critical path
    (MI355X guide: "capture launch-bound inner loops in hipGraphs").  Device-side RNG (drop-path masks, RoPE rescale) stays
    live under replay because torch registers the generator's Philox offset with the graph."""

    def __init__(self, net, optimizer, params, x_shape, tgt_shape, device, reducer=None, max_norm=12.0, graph=True, warmup=3,
                 ddp_loss=None, comm_outside_graph=None):
        self.net, self.opt, self.params, self.reducer, self.max_norm = net, optimizer, params, reducer, max_norm
        self.ddp_loss = ddp_loss         # None: batch-Dice sums all-reduced iff world size > 1 (dice.py:58-119 with ddp=True)
        if comm_outside_graph is None:
            comm_outside_graph = os.environ.get("DINOUNET_COMM_OUTSIDE_GRAPH", "0") == "1"
        self.comm_outside_graph = bool(comm_outside_graph) and reducer is not None
        self.capture_mode = "eager"      # -> "whole_step" | "segments(n)" once a capture exists (bench.py reports it)
        self._segments = None
        self.x = torch.zeros(x_shape, device=device)
        self.tgt = torch.zeros(tgt_shape, dtype=torch.long, device=device)
        self.loss = None
        self.graph = None
        self.use_graph = graph
        self.warmup = warmup
        self._n = 0
        self.comm_events = None          # list -> eager steps record (start, end) events around reducer.finish()
        self._opt_generation = 0

    def _step(self):
        self.opt.zero_grad(set_to_none=True)
        logits = self.net(self.x)
        loss = dc_and_ce_loss(logits, self.tgt, ddp=self.ddp_loss)
        loss.backward()
        if logits.is_cuda:
            from . import ops
            ops.WGRAD.flush()          # deferred weight gradients (already flushed by the engine callback; idempotent)
        if self.reducer is not None:
            if self._segments is not None:                  # segmented capture: every bucket's all-reduce from the host, between two graphs
                self.reducer.fill_missing()
                self._segments.cut(self.reducer.reduce_deferred, "gradient_buckets")
            if self.comm_events is not None:               # bench.py: how long the step waits for the gradient all-reduce after backward
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                self.reducer.finish()
                e1.record()
                self.comm_events.append((e0, e1))
            else:
                self.reducer.finish()
        if isinstance(self.opt, FusedClipSGD):              # clip + SGD in three launches (csrc/optim.hip)
            self.opt.max_norm = self.max_norm
            self.opt.step()
        else:
            torch.nn.utils.clip_grad_norm_(self.params, self.max_norm)
            self.opt.step()
        return loss.detach()

    def _forget_failed_pass(self):
        """a capture that raised part-way through backward: re-arm the reducer's bucket counters and drop the weight-gradient products
        the dead pass queued (nobody will read them; the queue would otherwise launch them at the next backward, ops.WgradQueue._arm)"""
        if self.reducer is not None:
            self.reducer.rearm()
        if self.x.is_cuda:
            from . import ops
            ops.WGRAD._drop()
            ops.ZEROS.rec = []           # the dead pass's truncated request record must not become the next step's plan

    def _all_ranks(self, ok):
        """the capture mode is a collective decision: MIN over the ranks of 'my capture worked' (outside any capture)"""
        if self.reducer is None or not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return ok
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.x.device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(int(t.item()))

    def _capture_in_segments(self, mode):
        import gc
        from . import ops
        seg = _CaptureSegments(mode)
        torch.cuda.synchronize()
        gc.collect()
        torch.cuda.empty_cache()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        self._segments, ops.CAPTURE_CUT, self.reducer.defer = seg, seg.cut, True
        try:
            with torch.cuda.stream(s):
                seg.begin()
                self.loss = self._step()
                seg.end()
        except BaseException:
            seg.abort()
            raise
        finally:
            self._segments, ops.CAPTURE_CUT, self.reducer.defer = None, None, False
        torch.cuda.current_stream().wait_stream(s)
        return seg

    def __call__(self, x=None, tgt=None):
        if x is not None:
            self.x.copy_(x, non_blocking=True)
        if tgt is not None:
            self.tgt.copy_(tgt, non_blocking=True)
        if not self.use_graph:
            self.loss = self._step()
            return self.loss
        if self.graph is None:
            if self._n < self.warmup:                  # eager warm-up on a side stream (allocator + lazy caches settle)
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    self.loss = self._step()
                torch.cuda.current_stream().wait_stream(s)
                self._n += 1
                return self.loss
            torch.cuda.synchronize()
            if dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl":
                # Quiesce RCCL's watchdog before the capture.  The eager warm-up steps left Work objects in its list; it retires them by
                # polling their events every ~100 ms from its own thread.  A poll that lands while this thread is beginning the capture is
                # the one window in which round 3 saw a watchdog-thread SIGABRT (1 of 18 single-rank runs, never reproduced): after the
                # synchronize above every one of those events is complete, so waiting out a few poll periods empties the list and the
                # watchdog has nothing to query during the capture (collectives issued INSIDE a capture are not handed to it).
                import time
                time.sleep(0.35)
            # With a process group alive, ProcessGroupNCCL's watchdog thread polls its work events (hipEventQuery) while this
            # thread captures; under the default "global" capture mode that call is illegal and the watchdog aborts the process
            # ("operation not permitted when stream is capturing").  thread_local confines the capture restrictions to this thread.
            mode = "thread_local" if (dist.is_available() and dist.is_initialized()) else "global"
            err = None
            if not self.comm_outside_graph:
                try:
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph, capture_error_mode=mode):
                        self.loss = self._step()
                    self.graph = graph
                    self.capture_mode = "whole_step"
                except Exception as e:  # noqa: BLE001
                    err = e
                    torch.cuda.synchronize()
                    self._forget_failed_pass()
                if not self._all_ranks(self.graph is not None) and self.graph is not None:
                    # another rank's capture failed: every rank leaves this mode together -- a rank replaying a whole-step graph issues
                    # its bucket all-reduces in hook-ready order, a segmented one in bucket-index order (mixed modes hang
                    # or sum the wrong buckets)
                    self.graph.reset()
                    self.graph = None
                    self.capture_mode = "eager"
                    err = RuntimeError("whole-step capture failed on another rank")
                    self._forget_failed_pass()
            if self.graph is None and self.reducer is not None:
                # the collectives outside the graphs: asked for, or the capture of the whole step (RCCL kernels recorded into it) failed.
                # First with the mode of the whole-step attempt (cuts from the capturing thread only); a step whose BACKWARD issues
                # collectives (SyncBatchNorm at N > 1: autograd's device thread) asks for a cut from that thread, which only a "relaxed"
                # capture may serve -- second attempt
                # (a capture that dies inside autograd's thread cannot be ended cleanly -- the graph's destructor throws and takes the
                #  process down, seen on hardware -- so the doomed thread-local attempt is not made when backward collectives are known to come)
                from . import ops as _ops
                for seg_mode in (("relaxed",) if _ops.sync_active() else (mode, "relaxed")):
                    try:
                        self.graph = self._capture_in_segments(seg_mode)
                        self.capture_mode = f"segments({len(self.graph.graphs)})" + ("" if seg_mode == mode else "/relaxed")
                        err = None
                    except Exception as e:  # noqa: BLE001
                        err = e
                        torch.cuda.synchronize()
                        self._forget_failed_pass()
                    if not self._all_ranks(self.graph is not None) and self.graph is not None:
                        self.graph.reset()
                        self.graph = None
                        self.capture_mode = "eager"
                        err = RuntimeError("segmented capture failed on another rank")
                        self._forget_failed_pass()
                    if self.graph is not None:
                        break
            if self.graph is None:     # capture is an optimisation: a step that cannot be captured still has to train
                import warnings
                warnings.warn(f"hipGraph capture of the train step failed ({err!r}); continuing with eager steps")
                self.use_graph = False
                torch.cuda.synchronize()
                self.loss = self._step()
                return self.loss
            self._opt_generation = getattr(self.opt, "generation", 0)
        if isinstance(self.opt, FusedClipSGD):
            if getattr(self.opt, "generation", 0) != self._opt_generation:
                # optimizer.load_state_dict / reallocated parameters since the capture: the recorded step points at retired tables and
                # momentum buffers -- drop it and capture again
                self.graph.reset()
                self.graph = None
                self._n = max(0, self.warmup - 1)
                return self.__call__()
            self.opt.refresh_hyper()                        # the captured step re-reads lr & co. from the pinned host buffer
        self.graph.replay()
        return self.loss
