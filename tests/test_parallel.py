"""World-size-2 gloo tests (CPU) of the data-parallel host logic in oct_segmentation_amd/parallel.py.

The HIP engine itself needs a GPU; what is testable here is the exchange protocol on the flat arenas
(broadcast at construction, SUM all-reduce of 1/W-scaled gradients == DDP's gradient mean, buffer
broadcast from rank 0) and that this protocol reproduces Lightning-DDP semantics for the oracle:
per-rank BN statistics + per-rank Dice, gradients averaged."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FakeNet:
    """Just the arenas parallel.py touches (CPU tensors)."""

    def __init__(self, n, seed):
        g = torch.Generator().manual_seed(seed)
        self.arena = torch.nn.Parameter(torch.randn(n, generator=g))
        self.bn_buffers = torch.randn(16, generator=g)
        self._grad_arena = torch.randn(n, generator=g)
        self.arena.grad = self._grad_arena


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from oct_segmentation_amd import parallel as P
        # --- arena protocol ---
        net = _FakeNet(1000, seed=rank)
        local_grad = net._grad_arena.clone()
        P.broadcast_parameters(net)
        ref0 = _FakeNet(1000, seed=0)
        assert torch.equal(net.arena.data, ref0.arena.data) and torch.equal(net.bn_buffers, ref0.bn_buffers)
        net._grad_arena.mul_(1.0 / world)          # what grad_scale = 1/W does inside the engine
        P.allreduce_gradients(net, world, average=False)
        expect = sum(_FakeNet(1000, seed=r)._grad_arena for r in range(world)) / world
        assert torch.allclose(net._grad_arena, expect, atol=1e-6)
        net._grad_arena.copy_(local_grad)
        P.allreduce_gradients(net, world, average=True)
        assert torch.allclose(net._grad_arena, expect, atol=1e-6)
        # gradients on the wire as bfloat16: against the fp32 exchange cosine >= 0.9999, error <= 2^-7 of the largest
        net._grad_arena.copy_(local_grad)
        P.allreduce_gradients(net, world, average=True, wire_dtype='bf16')
        cos = torch.nn.functional.cosine_similarity(net._grad_arena.double(), expect.double(), dim=0).item()
        assert cos >= 0.9999 and (net._grad_arena - expect).abs().max().item() <= 2.0 ** -7 * expect.abs().max().item() * world, cos
        net.bn_buffers.add_(rank + 1.0)
        P.broadcast_buffers(net)
        assert torch.equal(net.bn_buffers, ref0.bn_buffers + 1.0)
        lo, hi = P.shard_range(5, rank, world)
        assert (lo, hi) == ((0, 3) if rank == 0 else (3, 5))

        # --- DDP semantics on the oracle: shard the batch, local BN + local Dice, averaged gradients ---
        from oracle import create_model, DiceLoss
        from synth import make_batch
        torch.manual_seed(3)
        m = create_model('unet', 'resnet18', classes=1).train()
        img, mask = make_batch(4, 1, 32, seed=5)
        lo, hi = P.shard_range(4, rank, world)
        loss = DiceLoss()(m(img[lo:hi]), mask[lo:hi])
        loss.backward()
        flat = torch.cat([p.grad.flatten() for p in m.parameters()]) / world
        dist.all_reduce(flat)
        if rank == 0:
            out.put(flat)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_data_parallel_protocol_gloo_world2():
    ctx = mp.get_context('spawn')
    out = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    flat = out.get()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    # sequential emulation of the two ranks in this process
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from oracle import create_model, DiceLoss
    from synth import make_batch
    img, mask = make_batch(4, 1, 32, seed=5)
    acc = None
    for lo, hi in ((0, 2), (2, 4)):
        torch.manual_seed(3)
        m = create_model('unet', 'resnet18', classes=1).train()
        DiceLoss()(m(img[lo:hi]), mask[lo:hi]).backward()
        g = torch.cat([p.grad.flatten() for p in m.parameters()]) / 2
        acc = g if acc is None else acc + g
    assert torch.allclose(flat, acc, rtol=1e-5, atol=1e-7)
