"""Data parallelism: one process per GPU, ``torch.distributed`` backend "nccl" (= RCCL over xGMI on
ROCm; "gloo" in the CPU tests).  Semantics are those Lightning's DDPStrategy gives the reference
(``src/models/smp/train.py:122-133`` with ``devices='auto'``): every rank runs the full step on its
shard of the batch with local BN statistics and a local Dice loss, gradients are averaged, BN
buffers follow rank 0.

The parameters live in ONE flat fp32 arena, so the exchange is a single all-reduce of the gradient
arena (68 M elements = 272 MB for U-Net++/resnet101): no bucketing logic, no per-tensor launches,
and the ring runs at the per-link xGMI rate for its whole duration.
"""
import torch
import torch.distributed as dist


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


def allreduce_gradients(net, world=None, average=True):
    """Sum (and average) the flat gradient arena across ranks in one collective."""
    g = net.arena.grad if net.arena.grad is not None else net._grad_arena
    dist.all_reduce(g, op=dist.ReduceOp.SUM)
    if average:
        g.div_(world if world is not None else dist.get_world_size())
    return g
