"""CPU tests of the host protocol of ops.WgradQueue (deferred weight gradients, du_gemm_tn_group) with a stand-in for the library call:
what must hold whatever the kernels do -- p.grad is the very buffer the deferred launch fills (never a clone taken before the launch),
the queue is flushed before backward() returns, a parameter used several times in a pass gets ONE buffer, parameters that cannot be
deferred safely (existing gradient, hooks) are computed at once.  The kernels themselves are covered by the -m gpu tests."""
import ctypes as C

import pytest
import torch

from dinounet_amd import _lib, ops


class _FakeLib:
    """du_gemm_tn_group stand-in: writes (job.M + 0.5) into every element of job.C (fp32) -- `accumulate` jobs add instead -- at FLUSH time"""

    def __init__(self):
        self.launched = []

    def du_gemm_tn_group_legal(self, job):
        return 1

    def du_gemm_tn_group(self, arr, n, stream):
        for i in range(n):
            j = arr[i]
            cnt = j.M * j.N
            buf = (C.c_float * cnt).from_address(j.C)
            for e in range(cnt):
                buf[e] = (buf[e] if j.accumulate else 0.0) + j.M + 0.5
            self.launched.append((j.M, j.N, j.accumulate))
        return 0


@pytest.fixture
def queue(monkeypatch):
    fake = _FakeLib()
    monkeypatch.setattr(_lib, "lib", lambda: fake)
    monkeypatch.setattr(ops, "_st", lambda: None)
    q = ops.WgradQueue()
    q.enabled = True
    monkeypatch.setattr(ops, "WGRAD", q)
    ops.ZEROS.new_step()
    return q, fake


class _Lin(torch.autograd.Function):
    """y = x * sum(w): only the plumbing matters; the 'weight gradient' is whatever the (fake) deferred launch writes"""

    @staticmethod
    def forward(ctx, x, w):
        ctx.wrefs = ops.WGRAD.note_use(w)
        ctx.shape = tuple(w.shape)
        return x * w.sum()

    @staticmethod
    def backward(ctx, gy):
        grp = ops.WGRAD.begin(ctx.wrefs)
        if grp is None:
            return gy, torch.full(ctx.shape, -1.0)             # "computed at once"
        dw = ops.WGRAD.buffer(grp, "dw", ctx.shape, gy.device)
        job = _lib.TnJob(A=1, lda=8, B=1, ldb=8, C=dw.data_ptr(), ldc=ctx.shape[1], a_colsum=None, alpha=None, M=ctx.shape[0], N=ctx.shape[1],
                         K=512, accumulate=0)
        ops.WGRAD.add(job, (gy,), grp)
        return gy, (dw if grp["ret"] else None)


def test_gradient_is_the_deferred_buffer_and_complete_when_backward_returns(queue):
    q, fake = queue
    w1, w2 = torch.nn.Parameter(torch.ones(2, 3)), torch.nn.Parameter(torch.ones(4, 3))
    x = torch.ones(5, requires_grad=True)
    y = _Lin.apply(_Lin.apply(x, w1), w2)
    y.sum().backward()
    assert len(fake.launched) == 2 and not q.jobs and not q.keep and not q.state and not q.groups and q._armed_task is None
    # the values the launch wrote AFTER the nodes had returned their buffers: AccumulateGrad adopted the buffers, it did not clone them
    assert torch.equal(w1.grad, torch.full((2, 3), 2.5)) and torch.equal(w2.grad, torch.full((4, 3), 4.5))
    assert q.queued == 2 and q.launches == 1


def test_parameter_used_twice_in_a_pass_gets_one_buffer(queue):
    q, fake = queue
    w = torch.nn.Parameter(torch.ones(2, 3))
    x = torch.ones(5, requires_grad=True)
    _Lin.apply(_Lin.apply(x, w), w).sum().backward()
    assert [a for _, _, a in fake.launched] == [1, 1]          # both jobs accumulate into the one buffer (the first one was patched)
    assert torch.equal(w.grad, torch.full((2, 3), 5.0))        # 2.5 + 2.5, added by the launch, not by the autograd engine
    assert q.launches == 1


def test_existing_gradient_and_hooks_keep_a_product_out_of_the_queue(queue):
    q, fake = queue
    w = torch.nn.Parameter(torch.ones(2, 3))
    x = torch.ones(5, requires_grad=True)
    _Lin.apply(x, w).sum().backward()
    assert torch.equal(w.grad, torch.full((2, 3), 2.5))
    _Lin.apply(x, w).sum().backward()                          # p.grad exists: AccumulateGrad adds at once -> must be a complete tensor
    assert len(fake.launched) == 1 and torch.equal(w.grad, torch.full((2, 3), 1.5))
    w.grad = None
    seen = []
    h = w.register_hook(lambda g: seen.append(g.clone()))
    _Lin.apply(x, w).sum().backward()
    h.remove()
    assert len(fake.launched) == 1 and torch.equal(seen[0], torch.full((2, 3), -1.0))


def test_a_complete_contribution_flushes_what_is_pending_for_that_parameter(queue):
    """first use deferred, second use of the same weight cannot be (stand-in: a hook appears in between is not possible inside one pass, so
    the second node aborts its group as an illegal job would): the pending buffer must be complete before the engine adds the two"""
    q, fake = queue
    w = torch.nn.Parameter(torch.ones(2, 3))
    x = torch.ones(5, requires_grad=True)

    class _Abort(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w):
            ctx.wrefs = ops.WGRAD.note_use(w)
            return x * w.sum()

        @staticmethod
        def backward(ctx, gy):
            grp = ops.WGRAD.begin(ctx.wrefs)
            assert grp is not None and not grp["first"]
            ops.WGRAD.abort(grp)                               # job not legal -> computed at once
            assert len(fake.launched) == 1                     # ... after the first use's job has been launched
            return gy, torch.full((2, 3), 10.0)

    # backward order: the LAST applied function runs first -> _Lin (deferred) then _Abort
    _Lin.apply(_Abort.apply(x, w), w).sum().backward()
    assert torch.equal(w.grad, torch.full((2, 3), 12.5))


def test_a_backward_pass_that_raised_does_not_disarm_the_queue(queue):
    """the engine does not run final callbacks when a pass raises; a boolean 'armed' flag would stay set and every later
    backward() would return with unfinished gradients.  Arming is keyed on the graph task: the next pass launches what the dead one left
    queued (a nested pass looks the same from here and MUST NOT lose the outer pass's jobs, ADVICE r4 -- the leftovers of a dead pass go
    into buffers only the queue still holds) and arms itself."""
    q, fake = queue

    class _Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x * 1.0

        @staticmethod
        def backward(ctx, g):
            raise RuntimeError("boom")

    w = torch.nn.Parameter(torch.ones(2, 3))
    x = torch.ones(5, requires_grad=True)
    y = _Boom.apply(_Lin.apply(x, w))          # _Lin's backward runs AFTER _Boom's: nothing queued yet when it raises
    with pytest.raises(RuntimeError):
        y.sum().backward()
    y = _Lin.apply(_Boom.apply(x), w)          # _Lin queues its job, THEN the pass dies
    with pytest.raises(RuntimeError):
        y.sum().backward()
    assert q.jobs and q._armed_task is not None and not fake.launched      # the dead pass left its job behind
    w.grad = None
    w2 = torch.nn.Parameter(torch.ones(4, 3))
    _Lin.apply(x, w2).sum().backward()         # a healthy pass: the leftovers are launched out of the way, its own gradient is complete on return
    assert fake.launched == [(2, 3, 0), (4, 3, 0)] and not q.jobs and not q.keep and q._armed_task is None and not q.state and not q.groups
    assert torch.equal(w2.grad, torch.full((4, 3), 4.5)) and w.grad is None


def test_reset_after_a_failed_backward_drops_the_dead_jobs_instead_of_launching_them(queue):
    """with zero_grad(set_to_none=False) / gradient accumulation autograd may still hold a dead pass's result buffer as p.grad;
    launching the leftovers at the next backward would add a stale product into the new gradient.  WGRAD.reset() after the failure is the
    documented way out: nothing of the dead pass is launched, the next pass arms itself and completes its own gradients."""
    q, fake = queue

    class _Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x * 1.0

        @staticmethod
        def backward(ctx, g):
            raise RuntimeError("boom")

    w = torch.nn.Parameter(torch.ones(2, 3))
    x = torch.ones(5, requires_grad=True)
    with pytest.raises(RuntimeError):
        _Lin.apply(_Boom.apply(x), w).sum().backward()
    assert q.jobs and q._armed_task is not None
    q.reset()
    assert not q.jobs and not q.keep and not q.groups and not q.state and q._armed_task is None
    w2 = torch.nn.Parameter(torch.ones(4, 3))
    _Lin.apply(x, w2).sum().backward()
    assert fake.launched == [(4, 3, 0)]                       # the dead (2, 3) job was never launched
    assert torch.equal(w2.grad, torch.full((4, 3), 4.5))


def test_a_nested_backward_pass_does_not_lose_the_outer_passes_jobs(queue):
    """a backward pass started INSIDE a node of another pass (reentrant checkpointing, autograd.grad in a hook or a custom
    Function) has its own graph-task id.  The outer pass is alive and has already handed its zero-filled buffers to autograd: its queued
    jobs must be launched, not dropped, and both passes' gradients must be complete when the outer backward() returns."""
    q, fake = queue
    w_in = torch.nn.Parameter(torch.ones(6, 3))
    inner_grads = []

    class _Nested(torch.autograd.Function):
        """backward runs a whole inner pass (with a deferred product of its own) through autograd.grad"""

        @staticmethod
        def forward(ctx, x):
            return x * 1.0

        @staticmethod
        def backward(ctx, g):
            with torch.enable_grad():
                xi = torch.ones(5, requires_grad=True)
                yi = _Lin.apply(xi, w_in).sum()
            inner_grads.append(torch.autograd.grad(yi, [w_in])[0])
            return g

    w_out, w_out2 = torch.nn.Parameter(torch.ones(2, 3)), torch.nn.Parameter(torch.ones(4, 3))
    x = torch.ones(5, requires_grad=True)
    # backward order: last applied first -> _Lin(w_out) queues, THEN the nested pass runs, then _Lin(w_out2) queues in the outer pass again
    y = _Lin.apply(_Nested.apply(_Lin.apply(x, w_out2)), w_out)
    y.sum().backward()
    assert not q.jobs and not q.keep and q._armed_task is None and not q.state and not q.groups
    assert torch.equal(w_out.grad, torch.full((2, 3), 2.5)), w_out.grad            # the outer job queued BEFORE the nested pass survived it
    assert torch.equal(w_out2.grad, torch.full((4, 3), 4.5)), w_out2.grad          # ... and the outer pass re-armed itself afterwards
    assert len(inner_grads) == 1 and torch.equal(inner_grads[0], torch.full((6, 3), 6.5))
    assert sorted(fake.launched) == [(2, 3, 0), (4, 3, 0), (6, 3, 0)]


def test_slice_views_of_a_parameter_are_not_deferred(queue):
    """through a slice view SliceBackward would copy the still-zero buffer: only the Parameter itself or a same-size contiguous
    view of it may be deferred."""
    q, _ = queue
    w = torch.nn.Parameter(torch.ones(4, 3))
    assert q.note_use(w) is not None and q.note_use(w.view(2, 6)) is not None
    assert q.note_use(w[:2]) is None and q.note_use(w.t()) is None and q.note_use(w.expand(2, 4, 3)) is None


def test_a_buffer_autograd_did_not_adopt_is_added_after_the_launch(queue):
    """weight tying -- the parameter also has a non-queued use in the pass, so the engine adds the two contributions OUT OF
    PLACE as soon as both exist, i.e. while the deferred buffer is still zero.  flush() sees that p.grad is not the buffer and adds the
    product behind the launch."""
    q, fake = queue
    w = torch.nn.Parameter(torch.ones(2, 3))
    x = torch.ones(5, requires_grad=True)
    y = _Lin.apply(x, w).sum() + (w * 2.0).sum()          # deferred contribution 2.5 everywhere + plain contribution 2.0
    y.backward()
    assert torch.equal(w.grad, torch.full((2, 3), 4.5)), w.grad
