"""Data parallelism: one process per GPU, ``torch.distributed`` backend "nccl" (= RCCL over xGMI on
ROCm; "gloo" in the CPU tests).  Semantics are those Lightning's DDPStrategy gives the reference
(``src/models/smp/train.py:122-133`` with ``devices='auto'``): every rank runs the full step on its
shard of the batch with local BN statistics and a local Dice loss, gradients are averaged, BN
buffers follow rank 0.

The parameters live in ONE flat fp32 arena (68 M elements = 272 MB for U-Net++/resnet101), so a bucket is just a
contiguous range of it: ``GradientExchange`` asks the engine for ``nslices`` ranges (``octseg_net_backward_sliced``),
and each range is all-reduced on a communication stream the moment its last writer is enqueued -- the head / decoder
ranges travel over xGMI while the encoder half of the backward still computes (torch DDP's bucketed overlap, without
per-tensor hooks or copies into bucket buffers).  ``allreduce_gradients`` is the unsliced form (one collective after
the backward).
"""
import ctypes as C
import os
import sys

import torch
import torch.distributed as dist

from . import _lib as L


class GradientExchangeError(RuntimeError):
    """A collective of the step failed on this rank.  The process group has been ABORTED (peers blocked in a collective this rank
    never joined fail too instead of hanging): the process cannot take part in another step and should exit non-zero so that the
    launcher (torchrun) restarts the job from a checkpoint -- ``exit_on_exchange_failure`` does that."""


def abort_process_group():
    """Tear the default group down without waiting for outstanding collectives: ncclCommAbort (= RCCL's) through torch's
    ``_abort_process_group`` / ``ProcessGroupNCCL.abort``; ``destroy_process_group`` is NOT an abort -- it flushes outstanding work first
    and can itself block on a peer that is stuck.  gloo (CPU tests) has no abort: its group is destroyed."""
    if not dist.is_initialized():
        return 'not initialised'
    try:
        if dist.get_backend() == 'nccl':
            from torch.distributed import distributed_c10d as c10d
            if hasattr(c10d, '_abort_process_group'):
                c10d._abort_process_group()
                return 'aborted'
            c10d._get_default_group()._get_backend(torch.device('cuda')).abort()
            return 'aborted'
        dist.destroy_process_group()
        return 'destroyed'
    except Exception as e:   # noqa: BLE001 -- already on the failure path
        return f'abort failed: {e!r}'


def exit_on_exchange_failure(err, code=13):
    """What a training driver does with a GradientExchangeError: say so on stderr and leave with a non-zero code WITHOUT running
    interpreter shutdown (destructors of an aborted communicator can block) -- a fresh process is the only clean state."""
    print(f'[octseg] rank {os.environ.get("RANK", "0")}: {err}; exiting with code {code}', file=sys.stderr, flush=True)
    os._exit(code)


def _all_reduce_sum(t):
    """SUM all-reduce of a (CUDA) tensor on the default group.  "nccl" (= RCCL) reduces in place on the device; the CPU
    test backend "gloo" is fed through a host copy (independent of whether this torch build's gloo takes device tensors)."""
    if t.is_cuda and dist.get_backend() == 'gloo':
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        t.copy_(h)
        return None
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)


class GradientExchange:
    """Backward + gradient all-reduce of one data-parallel step, overlapped.

        ex = GradientExchange(net, nslices=3)
        loss, logits, stats = net.train_step_raw(img, mask, ..., grad_scale=1 / world, exchange=ex)

    On return the caller's stream waits for every collective: ``net.arena.grad`` holds the averaged gradients."""

    def __init__(self, net, nslices=3, wire_dtype='fp32'):
        """``wire_dtype='bf16'``: every slice is cast to bfloat16 on the communication stream, summed by the collective in bf16 and
        written back to the fp32 arena -- half the bytes on the per-link-bound xGMI ring (136 instead of 272 MB for U-Net++/resnet101,
        SURVEY section 5).  The cast runs beside the backward like the collective itself; the sum of W bf16 values carries ~3 significant
        digits, which the reference's fp32 DDP buckets do not lose: opt-in."""
        if wire_dtype not in ('fp32', 'bf16'):
            raise ValueError(f"wire_dtype must be 'fp32' or 'bf16', got {wire_dtype!r}")
        self.net, self.nslices, self.wire_dtype = net, int(nslices), wire_dtype
        self._wire = torch.empty(net.param_numel, dtype=torch.bfloat16, device=net.device) if wire_dtype == 'bf16' else None
        self.comm = torch.cuda.Stream(device=net.device)
        self.fired = []          # (slice, begin, end) in completion order of the last step (tests / diagnostics)
        self._works = []
        self._error = None
        self._cb = L.SLICE_CB(self._on_slice)   # keep the ctypes thunk alive

    def _on_slice(self, _user, k, begin, end):
        # called from C (ctypes swallows Python exceptions raised in a callback): keep the first one for backward() to re-raise
        try:
            self.fired.append((int(k), int(begin), int(end)))
            if self._error is not None:
                return            # a collective already failed: issue no more, peers are released by the abort below
            g = self.net._grad_arena[begin:end]
            with torch.cuda.stream(self.comm):
                if self._wire is not None:
                    h = self._wire[begin:end]      # persistent buffer: no allocator traffic across streams
                    h.copy_(g)
                    w = _all_reduce_sum(h)
                    if w is not None:
                        w.wait()                   # stream-ordered on comm (no host block): the copy back follows the collective
                    g.copy_(h)
                    w = None
                else:
                    w = _all_reduce_sum(g)
            if w is not None:
                self._works.append(w)
        except BaseException as e:   # noqa: BLE001 -- must not propagate into the C caller
            self._error = e

    def backward(self, plan, logits, target, grad_scale, generation=None):
        net = self.net
        if generation is not None and generation != plan.generation:
            raise RuntimeError(f'backward of a stale step: another forward of shape {plan.shape} ran on this network since the step '
                               f'whose gradients are being exchanged')
        self.fired, self._works, self._error = [], [], None
        cur = torch.cuda.current_stream(net.device)
        L.check(L.lib().octseg_net_backward_sliced(plan.handle, L.ptr(net.arena.data), L.ptr(net._grad_arena),
                                                   L.ptr(plan.ws(logits.device)), L.ptr(logits), L.ptr(target.contiguous()),
                                                   float(grad_scale), L.stream_ptr(), self.nslices,
                                                   C.c_void_p(self.comm.cuda_stream), self._cb, None))
        if self._error is not None:
            # some slices were reduced, some were not: this rank cannot finish the step and its peers may be blocked inside a
            # collective it never joined -- ABORT the communicator (not destroy_process_group, which waits for outstanding work and
            # can block on exactly those peers) so that they fail too instead of hanging; the caller must exit (GradientExchangeError)
            err = self._error
            how = abort_process_group()
            raise GradientExchangeError(f'gradient exchange failed in slice callback: {err!r} (process group {how})') from err
        for w in self._works:
            w.wait()              # nccl: the current stream waits for the collective (no host block)
        cur.wait_stream(self.comm)
        covered = sorted((b, e) for _, b, e in self.fired)
        if not covered or covered[0][0] != 0 or covered[-1][1] != net.param_numel or any(a[1] != b[0] for a, b in zip(covered, covered[1:])):
            raise RuntimeError(f'the slices reported by the engine do not tile the gradient arena: {covered}')
        return net._grad_arena


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of a global batch for ``rank`` (first ranks take the remainder)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_parameters(net, src=0):
    """Make every rank start from rank ``src``'s parameters and BN buffers (DDP construction)."""
    dist.broadcast(net.arena.data, src)
    dist.broadcast(net.bn_buffers, src)
    if hasattr(net, 'params_changed'):
        net.params_changed()


def broadcast_buffers(net, src=0):
    """torch DDP's default ``broadcast_buffers=True``: BN running stats follow rank ``src``."""
    dist.broadcast(net.bn_buffers, src)


def allreduce_gradients(net, world=None, average=True, wire_dtype='fp32'):
    """Sum (and average) the flat gradient arena across ranks in one collective."""
    g = net.arena.grad if net.arena.grad is not None else net._grad_arena
    if wire_dtype == 'bf16':
        h = g.to(torch.bfloat16)
        w = _all_reduce_sum(h)
        if w is not None:
            w.wait()
        g.copy_(h)
        w = None
    else:
        w = _all_reduce_sum(g)
    if w is not None:
        w.wait()
    if average:
        g.div_(world if world is not None else dist.get_world_size())
    return g
