"""Data-parallel gradient exchange for Dino U-Net on one MI355X node: one process per GPU, `torch.distributed`
("nccl" = RCCL over xGMI).  Replaces the torch DDP wrap of the reference trainer (nnUNetTrainer.py:216-218) for the
hot path:

  * only the trainable tensors (adapter / FAPM / decoder, <= 20 M elements for dinounet_l; the ViT is frozen) are reduced;
  * gradients are copied into a few large flat fp32 buckets in reverse registration order (decoder first, SPM last =
    the order autograd produces them), and each bucket's all-reduce is issued from a side HIP stream as soon as its
    last gradient has been accumulated, so communication overlaps the rest of backward;
  * xGMI is point-to-point (7 links x ~153 GB/s): an 80 MB ring all-reduce is ~1 ms, so a handful of >= 16 MB buckets
    keeps every collective bandwidth-bound rather than latency-bound;
  * parameters that never receive a gradient (decoder.seg_layers.{0,1} without deep supervision) are
    skipped, so there is no "unused parameter" hazard;
  * SUM then divide by world size == DDP's gradient averaging; grad-clip 12 runs afterwards on identical gradients;
  * like DDP's constructor, the reducer first broadcasts rank 0's parameters AND buffers (BatchNorm running statistics,
    num_batches_tracked) to every rank: nnU-Net does not seed ranks identically, so replicas would otherwise start -- and stay --
    different.
Works on CPU tensors with the gloo backend (used by the world_size-2 tests).
"""
import torch
import torch.distributed as dist


def broadcast_module_state(module, group=None, src=0):
    """Rank `src`'s parameters and buffers -> every rank, in flat chunks per dtype (what torch DDP does at construction,
    nnUNetTrainer.py:218).  Frozen backbone weights are included: every rank loads the same checkpoint, but a rank that failed to
    (or was seeded differently in a test) must not silently train against different features."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    by_dtype = {}
    seen = set()
    for t in list(module.parameters()) + list(module.buffers()):
        if t.data_ptr() in seen:
            continue
        seen.add(t.data_ptr())
        by_dtype.setdefault((t.dtype, t.device), []).append(t)
    with torch.no_grad():
        for (dt, dev), ts in by_dtype.items():
            chunk, n = [], 0
            def flush():
                if not chunk:
                    return
                flat = torch.cat([t.detach().reshape(-1) for t in chunk])
                dist.broadcast(flat, src=src, group=group)
                off = 0
                for t in chunk:
                    t.copy_(flat[off:off + t.numel()].view_as(t))
                    off += t.numel()
            for t in ts:
                chunk.append(t)
                n += t.numel()
                if n >= 64 * 1024 * 1024:
                    flush()
                    chunk, n = [], 0
            flush()


class GradAllReducer:
    def __init__(self, module, world_size=None, bucket_elems=4 * 1024 * 1024, group=None, skip=(), broadcast=True):
        self.group = group
        self.world = world_size or dist.get_world_size(group)
        if broadcast:
            broadcast_module_state(module, group)
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
        ds = getattr(getattr(module, "decoder", None), "deep_supervision", True)
        if not ds:
            n_seg = len(module.decoder.seg_layers)
            dead = tuple(f"decoder.seg_layers.{i}." for i in range(n_seg - 1))
            named = [(n, p) for n, p in named if not n.startswith(dead)]
        named = [(n, p) for n, p in named if not any(n.startswith(s) for s in skip)]
        named = named[::-1]
        self.buckets = []
        cur, cur_n = [], 0
        for n, p in named:
            cur.append((n, p))
            cur_n += p.numel()
            if cur_n >= bucket_elems:
                self.buckets.append(cur)
                cur, cur_n = [], 0
        if cur:
            self.buckets.append(cur)
        self.flat, self.views, self.pending, self.works, self.gather = [], [], [], [], []
        self.timeline = None         # list -> eager steps record, per bucket, (ready on the compute stream, all-reduce start, all-reduce done)
        self.defer = False           # True while TrainStep captures a step in segments: hooks gather, reduce_deferred() all-reduces (see there)
        self._hooks = []
        self._keep = []
        dev = named[0][1].device
        self.is_cuda = dev.type == "cuda"
        self.side = torch.cuda.Stream(device=dev) if self.is_cuda else None
        for bi, b in enumerate(self.buckets):
            total = sum(p.numel() for _, p in b)
            flat = torch.zeros(total, dtype=torch.float32, device=dev)
            views, off = [], 0
            for _, p in b:
                views.append(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
            self.flat.append(flat)
            self.views.append(views)
            self.pending.append(len(b))
            self.works.append(None)
            self.gather.append(self._gather_tables(b, views, dev) if self.is_cuda else None)
            for j, (_, p) in enumerate(b):
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi, j)))
        if self.is_cuda:
            from . import ops
            ops.WGRAD.flush_aware_hooks += 1       # these hooks flush the deferred weight gradients before they read a gradient

    @staticmethod
    def _gather_tables(bucket, views, dev):
        """One du_pack_weights launch (kind CAST, fp32 -> fp32) copies every gradient of a bucket into its slot of the flat buffer
        instead of one small copy per parameter (~110 launches of ~5 us per step for dinounet_l).  The source addresses change from
        step to step, so the descriptor table travels pinned host -> device each time; a hipGraph capture gets its own pinned copy
        (replays re-read the source of the captured memcpy; the captured gradient addresses, from the graph's pool, stay valid)."""
        n = len(bucket)
        host = torch.zeros((n, 8), dtype=torch.int64)
        pre = torch.zeros(n + 1, dtype=torch.int64)
        for i, ((_, p), v) in enumerate(zip(bucket, views)):
            host[i, 1] = v.data_ptr()
            host[i, 2] = 0 | (1 << 8)            # PK_CAST, fp32 destination
            host[i, 7] = p.numel()
            pre[i + 1] = pre[i] + (p.numel() + 4095) // 4096
        return {"host": host.pin_memory(), "host_cap": host.clone().pin_memory(), "dev": torch.zeros((n, 8), dtype=torch.int64, device=dev),
                "pre": pre.to(dev), "nblocks": int(pre[n]), "n": n}

    def _gather(self, bi):
        from . import _lib
        g = self.gather[bi]
        capturing = torch.cuda.is_current_stream_capturing()
        host = g["host_cap"] if capturing else g["host"]
        if not capturing and g.get("evt") is not None:
            g["evt"].synchronize()         # the previous step's asynchronous copy of this pinned table has been read (the host may run ahead)
        ptrs = []
        for _, p in self.buckets[bi]:
            gr = p.grad
            if gr.dtype != torch.float32 or not gr.is_contiguous():
                gr = gr.float().contiguous()
                self._keep.append(gr)
            ptrs.append(gr.data_ptr())
        host.numpy()[:, 0] = ptrs
        g["dev"].copy_(host, non_blocking=True)
        if not capturing:
            g["evt"] = torch.cuda.Event()
            g["evt"].record()
        _lib.check(_lib.lib().du_pack_weights(g["dev"].data_ptr(), g["pre"].data_ptr(), g["n"], g["nblocks"],
                                              torch.cuda.current_stream().cuda_stream), "du_pack_weights")

    def _make_hook(self, bi, j):
        def hook(p):
            if not self.is_cuda:
                self.views[bi][j].copy_(p.grad)
            self.pending[bi] -= 1
            if self.pending[bi] == 0:
                if self.is_cuda:
                    from . import ops
                    ops.WGRAD.flush()              # deferred weight-gradient products of this bucket (ops.WgradQueue) must exist now
                    self._gather(bi)
                self._launch(bi)
        return hook

    def _launch(self, bi):
        if self.defer:
            return
        flat = self.flat[bi]
        if self.is_cuda:
            rec = self.timeline is not None and not torch.cuda.is_current_stream_capturing()
            if rec:
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                ev[0].record()                         # the bucket is gathered (compute stream)
            self.side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.side):
                if rec:
                    ev[1].record()
                self.works[bi] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                if rec:
                    # ProcessGroupNCCL runs an asynchronous collective on its OWN internal stream: an event recorded on `side` right
                    # after the call marks the end of the enqueue, not of the all-reduce (the round-4 timeline showed start ->
                    # done = 13 us for a 30 MB bucket).  Work.wait() is a stream-level dependency (no host block): `side` then trails
                    # the collective and the "done" event sees its real end.  Only on the eager timeline steps: inside a hipGraph
                    # capture the extra cross-stream edge made capture_end() fault (round 5), and finish() waits for the work anyway.
                    self.works[bi].wait()
                    ev[2].record()
                    self.timeline.append((bi, ev))
        else:
            self.works[bi] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def reduce_deferred(self):
        """Every bucket's all-reduce issued from the host, all at once, after the backward pass: the collective leg of a step captured
        in segments (training.TrainStep, DINOUNET_COMM_OUTSIDE_GRAPH=1: graph | this | graph).  No collective is recorded into a
        hipGraph in that mode, at the price of the overlap with backward: the buckets queue on RCCL's stream back to back and the
        compute stream waits for the last one.  Stateless (no hook bookkeeping): it runs once per REPLAY, where no Python hook fires."""
        if self.is_cuda:
            self.side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.side):
                works = [dist.all_reduce(f, op=dist.ReduceOp.SUM, group=self.group, async_op=True) for f in self.flat]
            for w in works:
                w.wait()                 # stream-level: the compute stream trails each collective, the host does not block
        else:
            for w in [dist.all_reduce(f, op=dist.ReduceOp.SUM, group=self.group, async_op=True) for f in self.flat]:
                w.wait()

    def fill_missing(self):
        """Buckets some gradient of which never arrived (e.g. a frozen branch): reduce what there is."""
        if self.is_cuda:
            from . import ops
            ops.WGRAD.flush()
        for bi, b in enumerate(self.buckets):
            if self.pending[bi] != 0:
                for j, (_, p) in enumerate(b):
                    if p.grad is None:
                        self.views[bi][j].zero_()
                    elif self.is_cuda:
                        self.views[bi][j].copy_(p.grad)
                self.pending[bi] = 0
                self._launch(bi)

    def finish(self):
        """Wait for every bucket, install the averaged gradients (views of the flat buffers) as p.grad, re-arm."""
        self.fill_missing()
        for bi, b in enumerate(self.buckets):
            if self.works[bi] is not None:      # (None: deferred mode, reduce_deferred() has run / will run between the two graphs)
                self.works[bi].wait()
                if self.is_cuda:
                    torch.cuda.current_stream().wait_stream(self.side)
            self.flat[bi].div_(self.world)
            for j, (_, p) in enumerate(b):
                p.grad = self.views[bi][j]
            self.pending[bi] = len(b)
            self.works[bi] = None
        self._keep.clear()

    def rearm(self):
        """Forget a step that did not reach finish() (a capture that raised part-way through backward)."""
        for bi, b in enumerate(self.buckets):
            self.pending[bi] = len(b)
            self.works[bi] = None
        self._keep.clear()

    def remove(self):
        for h in self._hooks:
            h.remove()
        if self._hooks and self.is_cuda:
            from . import ops
            ops.WGRAD.flush_aware_hooks = max(0, ops.WGRAD.flush_aware_hooks - 1)
        self._hooks = []
