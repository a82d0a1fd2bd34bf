"""octseg_net_backward_sliced (the data-parallel backward: gradient-arena ranges handed out while the backward still runs; reference: torch DDP's
bucketed all-reduce under Lightning, src/models/smp/train.py:122-133) must report a range only once its LAST writer is enqueued.  The callback
here synchronises the device and snapshots the reported range; when the whole backward has run, every snapshot must equal the final arena --
for every op family that owns parameters (convs, BatchNorms, GroupNorm, depthwise convs, squeeze-excite excitations, PAN's one-channel pyramid),
since a range reported early would be all-reduced before its gradients exist."""
import ctypes as C

import pytest
import torch

from synth import make_batch

pytestmark = pytest.mark.gpu

CASES = [('unet', 'resnet18', 64), ('fpn', 'resnet18', 64), ('deeplabv3plus', 'resnet18', 64), ('unet', 'timm-regnety_120', 64), ('unet', 'efficientnet-b0', 64),
         ('manet', 'resnet34', 64), ('pan', 'resnet18', 128)]


@pytest.mark.parametrize('arch,enc,S', CASES, ids=lambda v: str(v))
@pytest.mark.parametrize('nslices', [3, 7])
def test_slices_are_final_when_reported(cuda, arch, enc, S, nslices):
    from oct_segmentation_amd import _lib as L
    from oct_segmentation_amd.engine import SegNet
    net = SegNet(arch, enc, classes=2, device=cuda, compute_dtype=torch.float32, seed=3).train()
    img, mask = (t.to(cuda) for t in make_batch(4, 2, S, seed=2))
    logits, loss, stats, plan = net._forward_loss(img, mask, True, [0.485, 0.456, 0.406], [0.229, 0.224, 0.225])
    comm = torch.cuda.Stream(device=cuda)
    snaps = {}

    def on_slice(_user, k, begin, end):
        torch.cuda.synchronize()
        snaps[int(k)] = (int(begin), int(end), net._grad_arena[begin:end].clone())
    cb = L.SLICE_CB(on_slice)
    L.check(L.lib().octseg_net_backward_sliced(plan.handle, L.ptr(net.arena.data), L.ptr(net._grad_arena), L.ptr(plan.ws(cuda)), L.ptr(logits), L.ptr(mask),
                                               1.0, L.stream_ptr(), nslices, C.c_void_p(comm.cuda_stream), cb, None))
    torch.cuda.synchronize()
    assert 1 <= len(snaps) <= nslices          # (a range that would hold no parameter -- one conv wider than arena / n -- is not reported)
    covered = sorted((b, e) for b, e, _ in snaps.values())
    assert covered[0][0] == 0 and covered[-1][1] == net.param_numel and all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
    for k, (b, e, snap) in snaps.items():
        now = net._grad_arena[b:e]
        assert torch.equal(snap, now), f'{arch}/{enc}: slice {k} [{b}, {e}) changed after it was reported: {int((snap != now).sum())} elements'
    assert float(net._grad_arena.abs().max()) > 0
