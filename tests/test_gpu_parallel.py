"""Data parallelism of the ENGINE (not just the host protocol, which tests/test_parallel.py covers on CPU).

Lightning-DDP semantics of the reference (src/models/smp/train.py:122-133, devices > 1): every rank
runs forward + Dice + backward on its shard with LOCAL BatchNorm statistics and a LOCAL Dice loss, gradients are averaged,
BN running buffers follow rank 0.  The expected numbers are committed fixtures produced by the CPU oracle emulating the
ranks one after the other (tests/golden/make_golden.py, DDP_CASES).

  * test_engine_reproduces_ddp_sharded_fixtures: the ranks emulated sequentially on one GPU (world 2 and 4).
  * test_two_ranks_one_gpu_real_process_group: two real processes on ONE GPU, torch.distributed over gloo, the engine's
    sliced backward (octseg_net_backward_sliced) with the all-reduce of every slice issued from its callback, fused optimizer
    step -- both ranks must hold the fixture's averaged gradients and identical parameters afterwards.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')
MEAN, STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]


def _grad_sums(names, grads):
    return (np.array([grads[n].double().sum().item() for n in names]), np.array([grads[n].double().abs().sum().item() for n in names]))


def _buffer_sums(sd):
    return np.array([[v.double().sum().item(), v.double().abs().sum().item()] for k, v in sd.items()
                     if k.endswith('running_mean') or k.endswith('running_var')])


def _check_against_fixture(g, gsum, gabs, losses, stats, rank0_bufs):
    rel = np.abs(gabs - g['grad_abs_sums']) / np.maximum(g['grad_abs_sums'], 1e-3 * g['grad_abs_sums'].max())
    assert rel.max() < 2e-3, f'sum|grad| deviates by {rel.max():.2e}'
    assert np.abs(gsum - g['grad_sums']).max() <= 2e-3 * g['grad_abs_sums'].max()
    assert np.abs(np.array(losses) - g['losses']).max() <= 1e-5
    if stats is not None:
        assert np.array_equal(stats, g['stats'])
    if rank0_bufs is not None:
        assert np.abs(rank0_bufs - g['rank0_buffer_sums']).max() <= 1e-4 * max(1.0, np.abs(g['rank0_buffer_sums']).max())


@pytest.mark.parametrize('name', ['ddp_unet_resnet18_w2', 'ddp_linknet_resnet50_w4'])
def test_engine_reproduces_ddp_sharded_fixtures(cuda, name):
    from golden.make_golden import DDP_CASES, build, case_batch, shard_range
    from oct_segmentation_amd.engine import SegNet
    arch, enc, classes, B, S, seed, world = DDP_CASES[name]
    g = np.load(os.path.join(GOLDEN, f'{name}.npz'))
    ref = build(arch, enc, classes, seed)
    img, mask = case_batch(B, classes, S, seed)
    net = SegNet(arch, enc, classes=classes, device=cuda, compute_dtype=torch.float32).train()
    acc, losses, stats, rank0_sd = None, [], [], None
    for r in range(world):
        net.load_state_dict(ref.state_dict())      # every rank starts from rank 0's parameters and buffers
        lo, hi = shard_range(B, r, world)
        loss, logits, st = net.train_step_raw(img[lo:hi].to(cuda), mask[lo:hi].to(cuda), normalize=True, mean=MEAN, std=STD,
                                              grad_scale=1.0 / world)
        torch.cuda.synchronize()
        acc = net._grad_arena.clone() if acc is None else acc + net._grad_arena
        losses.append(loss.item()); stats.append(st.cpu().numpy())
        if r == 0:
            rank0_sd = {k: v.cpu() for k, v in net.state_dict().items()}
    net._grad_arena.copy_(acc)
    net.arena.grad = net._grad_arena
    grads = {k: v.cpu() for k, v in net.named_grads().items()}
    gsum, gabs = _grad_sums([n for n, _ in ref.named_parameters()], grads)
    _check_against_fixture(g, gsum, gabs, losses, np.concatenate(stats, axis=0), _buffer_sums(rank0_sd))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from golden.make_golden import DDP_CASES, build, case_batch, shard_range
        from oct_segmentation_amd import parallel as P
        from oct_segmentation_amd.engine import SegNet
        from oct_segmentation_amd.model import FusedOptimizer
        arch, enc, classes, B, S, seed, _ = DDP_CASES['ddp_unet_resnet18_w2']
        torch.cuda.set_device(0)
        dev = torch.device('cuda:0')
        net = SegNet(arch, enc, classes=classes, device=dev, compute_dtype=torch.float32, seed=1000 + rank).train()   # ranks differ ...
        if rank == 0:
            net.load_state_dict(build(arch, enc, classes, seed).state_dict())
        P.broadcast_parameters(net)                                                                           # ... until DDP construction
        img, mask = case_batch(B, classes, S, seed)
        lo, hi = shard_range(B, rank, world)
        ex = P.GradientExchange(net, nslices=3)
        P.broadcast_buffers(net)
        loss, logits, st = net.train_step_raw(img[lo:hi].to(dev), mask[lo:hi].to(dev), normalize=True, mean=MEAN, std=STD,
                                              grad_scale=1.0 / world, exchange=ex)
        torch.cuda.synchronize()
        names = [n for n, _ in build(arch, enc, classes, seed).named_parameters()]
        gsum, gabs = _grad_sums(names, {k: v.cpu() for k, v in net.named_grads().items()})
        bufs = _buffer_sums({k: v.cpu() for k, v in net.state_dict().items()})
        garena = net._grad_arena.cpu()
        g0 = garena.clone()
        dist.broadcast(g0, 0)
        same_grads = bool(torch.equal(garena, g0))       # after the exchange EVERY rank holds the same averaged gradients
        # the same step with the gradients on the wire as bfloat16: cast per slice on the comm stream, bf16 SUM,
        # written back to the fp32 arena -- against the fp32 exchange: cosine >= 0.9999, Dice unchanged (the forward is the same)
        bn_keep = net.bn_buffers.clone()
        ex16 = P.GradientExchange(net, nslices=3, wire_dtype='bf16')
        loss16, _, _ = net.train_step_raw(img[lo:hi].to(dev), mask[lo:hi].to(dev), normalize=True, mean=MEAN, std=STD,
                                          grad_scale=1.0 / world, exchange=ex16)
        torch.cuda.synchronize()
        g16 = net._grad_arena.cpu()
        cos16 = float(torch.nn.functional.cosine_similarity(g16.double(), garena.double(), dim=0))
        rel16 = float((g16 - garena).abs().max() / garena.abs().max())
        g16o = g16.clone()
        dist.broadcast(g16o, 0)
        assert cos16 >= 0.9999 and rel16 <= 2.0 ** -7 and abs(loss16.item() - loss.item()) <= 1e-3, (cos16, rel16)
        assert torch.equal(g16, g16o), 'bf16 exchange: ranks disagree'
        assert len(ex16.fired) == 3
        net._grad_arena.copy_(garena.to(dev))
        net.bn_buffers.copy_(bn_keep)
        opt = FusedOptimizer(net, 'Adam', 1e-3, 1e-4)
        opt.step()
        torch.cuda.synchronize()
        arena = net.arena.data.cpu()
        other = arena.clone()
        dist.broadcast(other, 0)
        # plain numpy / python objects only: torch tensors travel through a queue as file descriptors that die with this process
        q.put((rank, float(loss.item()), st.cpu().numpy(), gsum, gabs, bufs, list(ex.fired), same_grads, bool(torch.equal(arena, other))))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_one_gpu_real_process_group(cuda):
    import torch.multiprocessing as mp
    from golden.make_golden import DDP_CASES, build
    arch, enc, classes, B, S, seed, world = DDP_CASES['ddp_unet_resnet18_w2']
    g = np.load(os.path.join(GOLDEN, 'ddp_unet_resnet18_w2.npz'))
    ctx = mp.get_context('spawn')
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    import time
    res = {}
    t0 = time.time()
    while len(res) < world:
        if not q.empty():
            item = q.get()
            res[item[0]] = item
            continue
        dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
        assert not dead, f'a rank died with exit code {dead}'
        assert time.time() - t0 < 400, 'ranks did not report'
        time.sleep(0.2)
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    losses = [res[r][1] for r in range(world)]
    stats = np.concatenate([res[r][2] for r in range(world)], axis=0)
    for r in range(world):   # after the exchange EVERY rank holds the averaged gradients
        _check_against_fixture(g, res[r][3], res[r][4], losses, stats, res[r][5] if r == 0 else None)
        fired = res[r][6]
        assert len(fired) == 3 and sorted(k for k, _, _ in fired) == [0, 1, 2]
        assert fired[0][0] == 2, f'the highest arena range (decoder / head) must complete first, got {fired}'
        assert res[r][7], 'gradients differ between the ranks after the exchange'
        assert res[r][8], 'parameters diverged between the ranks after the optimizer step'
    _ = build


_RCCL_WORLD1 = r'''
import json, os, sys
ROOT = sys.argv[1]; port = int(sys.argv[2]); mode = sys.argv[3] if len(sys.argv) > 3 else 'ok'
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dev = torch.device('cuda:0')
dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', world_size=1, rank=0, device_id=dev)
out = {}
try:
    from oct_segmentation_amd import _lib as L
    from oct_segmentation_amd import parallel as P
    from oct_segmentation_amd.engine import SegNet
    from oct_segmentation_amd.model import FusedOptimizer
    from synth import make_batch
    L.check(L.lib().octseg_set_deterministic(1))      # no racing atomics: the two backward runs below are comparable bit for bit
    net = SegNet('unet', 'resnet18', classes=2, device=dev, compute_dtype=torch.float32, seed=5).train()
    P.broadcast_parameters(net)                       # ncclBroadcast of the arena and of the BN buffers
    P.broadcast_buffers(net)
    img, mask = (t.to(dev) for t in make_batch(2, 2, 64, seed=8))
    loss0, _, _ = net.train_step_raw(img, mask, grad_scale=1.0)
    torch.cuda.synchronize()
    g0 = net._grad_arena.clone()
    net.bn_buffers.zero_()
    ex = P.GradientExchange(net, nslices=3)
    loss1, _, _ = net.train_step_raw(img, mask, grad_scale=1.0, exchange=ex)     # async_op all-reduce per slice on the comm stream
    torch.cuda.synchronize()
    out['backend'] = dist.get_backend()
    out['same_loss'] = bool(loss0.item() == loss1.item())
    out['same_grads'] = bool(torch.equal(g0, net._grad_arena))
    out['fired'] = [list(f) for f in ex.fired]
    out['works'] = len(ex._works)
    out['numel'] = int(net.param_numel)
    g2 = P.allreduce_gradients(net, world=1).clone()                              # the unsliced collective
    torch.cuda.synchronize()
    out['allreduce_identity'] = bool(torch.equal(g2, g0))
    before = net.arena.data.clone()
    FusedOptimizer(net, 'Adam', 1e-3, 0.0).step()
    torch.cuda.synchronize()
    out['stepped'] = bool(not torch.equal(before, net.arena.data) and torch.isfinite(net.arena.data).all())
    # gradients on the wire as bfloat16 through RCCL (ncclBfloat16 SUM): one rank = a round trip through bf16 rounding
    net2 = SegNet('unet', 'resnet18', classes=2, device=dev, compute_dtype=torch.float32, seed=5).train()
    ex16 = P.GradientExchange(net2, nslices=3, wire_dtype='bf16')
    net2.train_step_raw(img, mask, grad_scale=1.0, exchange=ex16)
    torch.cuda.synchronize()
    g16 = net2._grad_arena
    out['bf16_cos'] = float(torch.nn.functional.cosine_similarity(g16.double(), g0.double(), dim=0))
    out['bf16_roundtrip'] = bool(torch.equal(g16, g0.to(torch.bfloat16).float()))
    out['bf16_fired'] = len(ex16.fired)
    if mode == 'fail':
        # a collective that raises inside a slice callback: the step must surface GradientExchangeError with the communicator ABORTED
        # (not destroy_process_group, which can block on peers), and the driver's handler must leave with its non-zero code
        real = P._all_reduce_sum
        calls = []
        def boom(t):
            calls.append(1)
            if len(calls) == 2:
                raise RuntimeError('injected collective failure')
            return real(t)
        P._all_reduce_sum = boom
        try:
            net.train_step_raw(img, mask, grad_scale=1.0, exchange=ex)
            out['raised'] = False
        except P.GradientExchangeError as e:
            out['raised'] = True
            out['message'] = str(e)
            out['group_alive'] = bool(dist.is_initialized())
            print('RESULT ' + json.dumps(out), flush=True)
            P.exit_on_exchange_failure(e, code=13)
finally:
    if dist.is_initialized():
        dist.destroy_process_group()
print('RESULT ' + json.dumps(out), flush=True)
'''


@pytest.mark.timeout(600)
def test_rccl_branch_world_size_one(cuda):
    """The RCCL ("nccl") branch of the exchange, executed for real on the one GPU there is: init_process_group('nccl', world_size=1,
    device_id=...), GradientExchange(nslices=3) forced on -- async_op all-reduces issued from the slice callbacks on the communication
    stream, Work.wait() and the stream joins -- must leave the gradient arena bit-identical to the exchange-free step (an all-reduce
    over one rank is the identity), with the slices fired decoder / head range first (reference: torch DDP's bucket order under
    Lightning, src/models/smp/train.py:122-133).  Runs in a child process so that the process group never leaks into other tests."""
    import json
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, '-c', _RCCL_WORLD1, ROOT, str(_free_port())], capture_output=True, text=True, timeout=540, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('RESULT ')][-1]
    out = json.loads(line[len('RESULT '):])
    assert out['backend'] == 'nccl' and out['works'] == 3
    assert out['same_loss'] and out['same_grads'] and out['allreduce_identity'] and out['stepped'], out
    fired = out['fired']
    assert sorted(k for k, _, _ in fired) == [0, 1, 2] and [k for k, _, _ in fired] == [2, 1, 0], fired
    assert fired[-1][1] == 0 and fired[0][2] == out['numel']
    assert out['bf16_fired'] == 3 and out['bf16_roundtrip'] and out['bf16_cos'] >= 0.9999, out


@pytest.mark.timeout(600)
def test_failed_collective_aborts_the_group_and_exits_nonzero(cuda):
    """a collective that fails inside a slice callback must not leave the rank continuing in-process on a half-reduced
    gradient arena, and must not call destroy_process_group (which flushes outstanding work and can block on the peers): the RCCL
    communicator is aborted (torch's _abort_process_group -> ncclCommAbort), GradientExchangeError surfaces, and the driver's handler
    (parallel.exit_on_exchange_failure, what fit() and bench.py install) leaves with exit code 13."""
    import json
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, '-c', _RCCL_WORLD1, ROOT, str(_free_port()), 'fail'], capture_output=True, text=True, timeout=540, env=env)
    assert r.returncode == 13, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('RESULT ')][-1][len('RESULT '):])
    assert out['raised'] and 'injected collective failure' in out['message'] and 'aborted' in out['message'], out
    assert 'exiting with code 13' in r.stderr
