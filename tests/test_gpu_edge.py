"""Edge cases of the hot path on the GPU: non-square inputs, batch 1, four classes, every class empty,
plan reuse across shapes, eval/train switching, bf16 full-size property checks (size-independent
invariants at BASELINE.json's 704x704), and the thin train loop with checkpoint hand-over to predict."""
import json
import os

import numpy as np
import pytest
import torch

from synth import make_batch

pytestmark = pytest.mark.gpu


def _pair(arch, enc, classes, cuda, dtype=torch.float32):
    from oct_segmentation_amd.engine import SegNet
    from test_gpu_net import _oracle
    ref = _oracle(arch, enc, classes, kinkfree=True)
    net = SegNet(arch, enc, classes=classes, device=cuda, compute_dtype=dtype)
    net.load_state_dict(ref.state_dict())
    return ref, net


@pytest.mark.parametrize('arch,classes,B,H,W', [('unet', 1, 1, 96, 64), ('linknet', 4, 3, 64, 128), ('unetplusplus', 3, 1, 64, 32)])
def test_nonsquare_batch1_multiclass(cuda, arch, classes, B, H, W):
    from oracle import DiceLoss, get_stats
    from test_gpu_net import _grad_report
    ref, net = _pair(arch, 'resnet18', classes, cuda)
    g = torch.Generator().manual_seed(3)
    img = (torch.rand(B, 3, H, W, generator=g) * 255).round()
    mask = (torch.rand(B, classes, H, W, generator=g) > 0.6).float()
    ref.train(); net.train()
    z = ref(img)
    loss_ref = DiceLoss()(z, mask)
    loss_ref.backward()
    loss, logits, stats = net.train_step_raw(img.to(cuda), mask.to(cuda))
    torch.cuda.synchronize()
    scale = z.detach().abs().max().item()
    assert (logits.cpu() - z.detach()).abs().max().item() <= 2e-4 * max(1.0, scale)
    assert abs(loss.item() - loss_ref.item()) <= 1e-5
    tp, fp, fn, tn = get_stats((logits.cpu().sigmoid() > 0.5).long(), mask.long())
    assert torch.equal(stats.cpu(), torch.stack([tp, fp, fn, tn], dim=-1))
    cos, worst, name = _grad_report(net.named_grads(), ref)
    # batch 1 at 64x32 leaves 2 samples per channel in layer4: BN backward is ill-conditioned there (and the split-K
    # atomics order varies run to run), hence 1e-2 on the worst element beside the cosine
    assert cos > 0.999999 and worst < 1e-2, (cos, worst, name)


def test_single_value_per_channel_raises_like_torch(cuda):
    from oct_segmentation_amd.engine import SegNet
    net = SegNet('unet', 'resnet18', classes=1, device=cuda, compute_dtype=torch.float32).train()
    with pytest.raises(ValueError, match='Expected more than 1 value per channel when training'):
        net(torch.zeros(1, 3, 32, 32, device=cuda))      # layer4 is 1x1 with batch 1
    net.eval()
    assert net(torch.zeros(1, 3, 32, 32, device=cuda)).shape == (1, 1, 32, 32)


def test_all_classes_empty_gives_zero_loss_and_zero_grads(cuda):
    _, net = _pair('unet', 'resnet18', 2, cuda)
    img, _ = make_batch(2, 2, 64, seed=1)
    net.train()
    loss, logits, stats = net.train_step_raw(img.to(cuda), torch.zeros(2, 2, 64, 64, device=cuda))
    torch.cuda.synchronize()
    assert loss.item() == 0.0
    assert float(net.arena.grad.abs().max()) == 0.0
    assert int(stats[..., 0].sum()) == 0 and int(stats[..., 2].sum()) == 0     # tp = fn = 0


def test_plan_cache_and_mode_switch(cuda):
    ref, net = _pair('unet', 'resnet18', 1, cuda)
    ref.eval(); net.eval()
    for (B, S) in ((1, 64), (2, 96), (1, 64)):
        img, _ = make_batch(B, 1, S, seed=B + S)
        with torch.no_grad():
            z = ref(img)
        y = net(img.to(cuda)).cpu()
        assert (y - z).abs().max().item() <= 1e-4 * max(1.0, z.abs().max().item())
    assert len(net._plans) == 2
    # train-mode forward updates the running statistics exactly once per call
    before = net.bn_buffers.clone()
    net.train()
    img, mask = make_batch(2, 1, 64, seed=9)
    net(img.to(cuda))
    assert int(net.num_batches_tracked) == 1 and not torch.equal(before, net.bn_buffers)
    with pytest.raises(RuntimeError, match='divisible by 32'):
        net(torch.zeros(1, 3, 48, 64, device=cuda))
    with pytest.raises(ValueError):
        net(torch.zeros(1, 1, 64, 64, device=cuda))


def test_eval_on_one_plan_sees_running_stats_updated_through_another(cuda):
    """Eval weight images fold the running statistics in and are cached per (B, H, W) plan.  A train-mode forward on plan B rewrites
    bn_buffers through the raw pointer (no torch version bump, no optimizer step): an eval forward on plan A afterwards must fold the
    NEW statistics (BN recalibration / training_step without opt.step followed by predict)."""
    ref, net = _pair('unet', 'resnet18', 1, cuda)
    imgA, _ = make_batch(1, 1, 64, seed=5)
    imgB, _ = make_batch(2, 1, 96, seed=6)
    ref.eval(); net.eval()
    with torch.no_grad():
        z0 = ref(imgA)
    y0 = net(imgA.to(cuda)).cpu()                     # plan A packed for eval with the initial statistics
    assert (y0 - z0).abs().max().item() <= 1e-4 * max(1.0, z0.abs().max().item())
    ref.train(); net.train()
    with torch.no_grad():
        ref(imgB)                                     # updates the oracle's running statistics
    net(imgB.to(cuda))                                # plan B, train mode: bn_buffers rewritten in place
    ref.eval(); net.eval()
    with torch.no_grad():
        z1 = ref(imgA)
    y1 = net(imgA.to(cuda)).cpu()                     # plan A again: must not reuse the stale folded images
    assert (z1 - z0).abs().max().item() > 1e-3        # (the statistics really moved the eval output)
    assert (y1 - z1).abs().max().item() <= 1e-4 * max(1.0, z1.abs().max().item())


def test_full_size_properties_bf16_704(cuda):
    """BASELINE size (704x704, bf16), U-Net/resnet18 to keep the CPU side out of it: properties that do not
    need the oracle -- finiteness, determinism of the forward, batch-permutation equivariance in eval mode,
    confusion counts consistent with the logits, gradient scaling linear in grad_scale."""
    from oct_segmentation_amd.engine import SegNet
    net = SegNet('unet', 'resnet18', classes=2, device=cuda, compute_dtype=torch.bfloat16, seed=3)
    img, mask = make_batch(4, 2, 704, seed=21)
    img, mask = img.to(cuda), mask.to(cuda)
    net.eval()
    y1 = net(img)
    y2 = net(img)
    assert torch.isfinite(y1).all() and torch.equal(y1, y2)
    perm = torch.tensor([2, 0, 3, 1], device=cuda)
    assert torch.equal(net(img[perm]), y1[perm])
    net.train()
    loss, logits, stats = net.train_step_raw(img, mask, grad_scale=1.0)
    g1 = net.arena.grad.clone()
    s = stats.cpu()
    pred = (logits.sigmoid() > 0.5)
    assert int(s[..., 0].sum()) == int((pred & (mask > 0)).sum())
    assert torch.equal(s.sum(-1), torch.full_like(s[..., 0], 704 * 704))
    net.bn_buffers.zero_()
    loss2, _, _ = net.train_step_raw(img, mask, grad_scale=0.5)
    assert abs(loss2.item() - loss.item()) < 1e-6
    ratio = (net.arena.grad.norm() / g1.norm()).item()
    assert abs(ratio - 0.5) < 2e-2      # bf16 dL/dlogits rounding + atomics order


def test_fit_loop_writes_reference_style_model_dir(cuda, tmp_path):
    from oct_segmentation_amd.config import load_config
    from oct_segmentation_amd.predict import load_model
    from oct_segmentation_amd.train import fit
    cfg = load_config('train', ['architecture=unet', 'encoder=resnet18', 'epochs=2', 'input_size=64', 'batch_size=2', 'lr=0.001',
                                'compute_dtype=fp32', 'use_augmentation=false'])
    cfg['classes'] = ['Lumen']
    batches = [tuple(t.to(cuda) for t in make_batch(2, 1, 64, seed=s)) for s in (1, 2, 3)]
    model, hist = fit(cfg, batches, val_batches=batches[:1], device=cuda, model_dir=str(tmp_path))
    assert len(hist) == 2 and 'train' in hist[0] and 'test' in hist[0]
    assert float(hist[1]['train']['loss']) < float(hist[0]['train']['loss']) + 1e-3   # it learns (or at least does not diverge)
    with open(os.path.join(tmp_path, 'config.json')) as f:
        j = json.load(f)
    assert set(j) == {'model_name', 'architecture', 'encoder', 'input_size', 'classes', 'batch_size', 'optimizer', 'lr'}
    import csv
    rows = list(csv.DictReader(open(os.path.join(tmp_path, 'metrics.csv'))))   # train + test rows, per class + Mean, per epoch
    assert len(rows) == 2 * 2 * 2 and {r['Split'] for r in rows} == {'train', 'test'} and rows[1]['Class'] == 'Mean'
    # the reference's order: validation ('test') rows of an epoch precede its training rows (model.py:134-148 before :97-106;
    # eval/training/Lumen/fold_1/metrics.csv:2-3)
    assert [(r['Epoch'], r['Split']) for r in rows] == [('1', 'test')] * 2 + [('1', 'train')] * 2 + [('2', 'test')] * 2 + [('2', 'train')] * 2
    m2, cfg2 = load_model(str(tmp_path), 'cuda', torch.float32)
    assert torch.equal(m2.model.arena.data, model.model.arena.data)
    out = m2.predict(np.zeros((1, 64, 64, 3), np.float32), 'cuda')
    assert out.shape == (1, 64, 64, 1)


def test_fit_with_deferred_metrics_writes_the_same_csv(cuda, tmp_path):
    """cfg['defer_metrics']: counts and losses stay on the GPU, one host copy per epoch (the reference syncs every step,
    utils.py:25-35).  Same steps (deterministic reductions), so metrics.csv must be byte-identical to the per-step path."""
    from oct_segmentation_amd import _lib as L
    from oct_segmentation_amd.config import load_config
    from oct_segmentation_amd.train import fit
    cfg = load_config('train', ['architecture=linknet', 'encoder=resnet18', 'epochs=2', 'input_size=64', 'batch_size=2', 'lr=0.001',
                                'compute_dtype=fp32', 'use_augmentation=false'])
    cfg['classes'] = ['Lipid core', 'Fibrous cap']
    batches = [tuple(t.to(cuda) for t in make_batch(2, 2, 64, seed=s)) for s in (4, 5, 6)]
    L.check(L.lib().octseg_set_deterministic(1))
    try:
        out = []
        for defer in (False, True):
            torch.manual_seed(7)
            d = tmp_path / ('defer' if defer else 'step')
            model, hist = fit(dict(cfg, defer_metrics=defer), batches, val_batches=batches[1:], device=cuda, model_dir=str(d))
            assert model.defer_metrics is defer
            out.append(((d / 'metrics.csv').read_bytes(), hist))
    finally:
        L.check(L.lib().octseg_set_deterministic(0))
    assert out[0][0] == out[1][0] and len(out[0][0]) > 200
    assert out[0][1][1]['val/loss'] == out[1][1][1]['val/loss']


def test_fit_loop_with_gpu_augmentation(cuda, tmp_path):
    """train.yaml's use_augmentation=true (the reference default) routes every batch through octseg_augment."""
    from oct_segmentation_amd.config import load_config
    from oct_segmentation_amd.train import fit
    cfg = load_config('train', ['architecture=linknet', 'encoder=resnet18', 'epochs=2', 'input_size=64', 'batch_size=4', 'lr=0.001'])
    assert cfg['use_augmentation'] is True
    cfg['classes'] = ['Lipid core', 'Fibrous cap']
    batches = [tuple(t.to(cuda) for t in make_batch(4, 2, 64, seed=s)) for s in (1, 2)]
    model, hist = fit(cfg, batches, device=cuda, augment_seed=3)
    assert len(hist) == 2 and np.isfinite(float(hist[1]['train']['loss'])) and 0.0 <= float(hist[1]['train']['loss']) <= 1.0
    from oct_segmentation_amd import augment as A
    assert np.array_equal(A.sample_params(4, 64, np.random.default_rng(3)), A.sample_params(4, 64, np.random.default_rng(3)))


def test_dice_gradient_of_saturated_logits(cuda):
    """Regression of round 1's fuzz case k=19 (cosine 0.99899): with |logits| > 17 the sigmoid rounds to 1 in fp32 and dp/dz
    written as p * (1 - p) vanished for every confidently-positive pixel, while autograd of logsigmoid(z).exp() keeps
    exp(-z).  A head scaled x40 saturates the net; every gradient must still match the oracle."""
    from oracle import DiceLoss
    from test_gpu_net import _grad_report
    ref, net = _pair('unet', 'resnet18', 1, cuda)
    img, mask = make_batch(3, 1, 64, seed=19)
    ref.train(); net.train()
    with torch.no_grad():
        z0 = ref(img)
        lo, hi = z0.quantile(0.2), z0.quantile(0.8)       # logits of both signs: the 20 % / 80 % quantiles land on -25 / +25
        k = 50.0 / (hi - lo)
        ref.segmentation_head[0].bias.sub_((hi + lo) / 2)
        ref.segmentation_head[0].weight.mul_(k)
        ref.segmentation_head[0].bias.mul_(k)
    net.load_state_dict(ref.state_dict())
    z = ref(img)
    assert float((z > 17).float().mean()) > 0.1 and float((z < -17).float().mean()) > 0.1, 'the case must saturate both ways'
    DiceLoss()(z, mask).backward()
    loss, logits, _ = net.train_step_raw(img.to(cuda), mask.to(cuda))
    torch.cuda.synchronize()
    cos, worst, name = _grad_report(net.named_grads(), ref)
    print(f'saturated head: |z| max {z.abs().max().item():.1f}, grad cosine {cos:.8f}, worst {worst:.2e} ({name})')
    assert cos > 0.999999 and worst < 5e-3


def test_backward_of_a_stale_step_raises(cuda):
    """One workspace per (B, H, W) plan holds the activations saved for backward: another forward of that shape between
    training_step and loss.backward() overwrites them.  The reference (autograd keeps every graph alive) would still give the right
    gradient; the engine cannot, so it must refuse instead of returning a wrong one."""
    from oct_segmentation_amd.model import OCTSegmentationModel
    m = OCTSegmentationModel('unet', 'resnet18', 'x', 3, ['Lumen'], device=cuda, compute_dtype=torch.float32, seed=2).train()
    img, mask = (t.to(cuda) for t in make_batch(2, 1, 64, seed=3))
    out = m.training_step((img, mask), 0)
    with torch.no_grad():
        m.eval(); m(img); m.train()                   # e.g. a validation forward of the same shape
    with pytest.raises(RuntimeError, match='stale step'):
        out['loss'].backward()
    out = m.training_step((img, mask), 0)             # the ordinary order still works
    out['loss'].backward()
    assert m.model.arena.grad is not None and torch.isfinite(m.model.arena.grad).all()
    # a forward of ANOTHER shape does not touch this plan
    out = m.training_step((img, mask), 0)
    with torch.no_grad():
        m.eval(); m(img[:1]); m.train()
    out['loss'].backward()


def test_encoder_weights_and_smp_kwargs(cuda, tmp_path):
    """smp.create_model's encoder_weights (reference default 'imagenet', model.py:38-44): a torchvision ResNet state_dict / file is
    loaded into encoder.*; 'imagenet' without a local file raises instead of silently training from scratch; unknown or
    non-default smp keywords are rejected instead of dropped."""
    from oct_segmentation_amd.engine import SegNet, create_model
    from oracle import create_model as oracle_model
    torch.manual_seed(3)
    enc = oracle_model('unet', 'resnet18', classes=1).encoder
    tv = {k: v.clone() for k, v in enc.state_dict().items()}
    tv['fc.weight'] = torch.zeros(1000, 512); tv['fc.bias'] = torch.zeros(1000)     # torchvision files carry the classifier too
    path = os.path.join(tmp_path, 'resnet18.pth')
    torch.save(tv, path)
    for src in (tv, path):
        net = SegNet('linknet', 'resnet18', encoder_weights=src, classes=2, device=cuda, compute_dtype=torch.float32, seed=9)
        sd = net.state_dict()
        for k, v in enc.state_dict().items():
            assert torch.equal(sd['encoder.' + k].cpu(), v), k
    with pytest.raises(RuntimeError, match='no pretrained weights can be downloaded'):
        SegNet('unet', 'resnet18', encoder_weights='imagenet', device=cuda)
    os.environ['OCTSEG_IMAGENET_DIR'] = str(tmp_path)
    try:
        net = create_model('unet', 'resnet18', encoder_weights='imagenet', classes=1, device=cuda, compute_dtype=torch.float32)
        assert torch.equal(net.state_dict()['encoder.conv1.weight'].cpu(), tv['conv1.weight'])
    finally:
        del os.environ['OCTSEG_IMAGENET_DIR']
    with pytest.raises(RuntimeError, match='do not fit'):
        SegNet('unet', 'resnet34', encoder_weights=tv, device=cuda)
    with pytest.raises(NotImplementedError):
        SegNet('unet', 'resnet18', device=cuda, decoder_attention_type='scse')
    with pytest.raises(TypeError):
        SegNet('unet', 'resnet18', device=cuda, not_an_smp_option=1)
    SegNet('unet', 'resnet18', device=cuda, decoder_use_batchnorm=True, encoder_depth=5, decoder_channels=[256, 128, 64, 32, 16])


@pytest.mark.parametrize('arch,enc', [('unetplusplus', 'resnet34'), ('linknet', 'resnet50')])
def test_deterministic_mode_gives_bit_identical_steps(cuda, arch, enc):
    """octseg_set_deterministic(1): the same step twice -> the same loss, logits, confusion counts, BN buffers and GRADIENT ARENA bit
    for bit (no split-K / Dice / bias atomics racing), and it agrees with the default mode to rounding."""
    from oct_segmentation_amd import _lib as L
    from oct_segmentation_amd.engine import SegNet
    net = SegNet(arch, enc, classes=2, device=cuda, compute_dtype=torch.float32, seed=6).train()
    img, mask = (t.to(cuda) for t in make_batch(3, 2, 160, seed=8))
    buf0 = net.bn_buffers.clone()

    def run():
        net.bn_buffers.copy_(buf0)
        loss, logits, stats = net.train_step_raw(img, mask)
        torch.cuda.synchronize()
        return loss.clone(), logits.clone(), stats.clone(), net._grad_arena.clone(), net.bn_buffers.clone()
    ref = run()
    L.check(L.lib().octseg_set_deterministic(1))
    try:
        a, b = run(), run()
    finally:
        L.check(L.lib().octseg_set_deterministic(0))
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert torch.equal(a[1], ref[1]) and torch.equal(a[2], ref[2])
    scale = ref[3].abs().max().item()
    assert (a[3] - ref[3]).abs().max().item() <= 1e-4 * scale and abs(a[0].item() - ref[0].item()) <= 1e-6   # another fp32 summation order


def test_reference_attribute_paths(cuda):
    """``model.model.encoder.layer4[-1]`` exists (reference src/models/visualize_activation_maps.py:103) as a named handle; hooks are
    refused loudly rather than silently recording nothing."""
    from oct_segmentation_amd.model import OCTSegmentationModel
    m = OCTSegmentationModel('unet', 'resnet50', 'x', 3, ['Lumen'], device=cuda, compute_dtype=torch.float32, seed=1)
    last = m.model.encoder.layer4[-1]
    assert len(m.model.encoder.layer4) == 3 and repr(last).endswith("['bn1', 'bn2', 'bn3', 'conv1', 'conv2', 'conv3']>")
    w = dict(last.ref().named_parameters())['conv3.weight']
    assert tuple(w.shape) == (2048, 512, 1, 1) and torch.equal(w, m.model.state_dict()['encoder.layer4.2.conv3.weight'])
    assert len(m.model.decoder.blocks) == 5
    with pytest.raises(NotImplementedError, match='no per-layer activations'):
        last.register_forward_hook(lambda *a: None)
    with pytest.raises(AttributeError):
        m.model.encoder.layer5


@pytest.mark.parametrize('arch,enc,dtype', [('unetplusplus', 'resnet18', torch.float32), ('linknet', 'resnet50', torch.bfloat16), ('fpn', 'resnet18', torch.float32),
                                            ('deeplabv3plus', 'resnet18', torch.float32),
                                            # round 4: depthwise / swish / squeeze-excite / drop_connect sweeps, MAnet's attention block, RegNet's per-group convs
                                            ('unet', 'efficientnet-b0', torch.float32), ('manet', 'resnet18', torch.bfloat16), ('unet', 'timm-regnety_120', torch.bfloat16)])
def test_captured_training_step_equals_eager_bit_for_bit(cuda, arch, enc, dtype):
    """octseg_net_train_step under octseg_plan_set_train_graph: warm-up call, capture, replay -- the replayed hipGraph (weight packing,
    forward lanes, the weight-gradient side stream and every event edge inside) must leave loss, logits, confusion counts, BatchNorm
    running statistics and the whole gradient arena bit-identical to the eager launches (deterministic reductions on both sides), and a
    second replay on NEW parameters must follow them (reference: training_step + loss.backward(), model.py:73-95, every step)."""
    from oct_segmentation_amd import _lib as L
    from oct_segmentation_amd.engine import SegNet
    classes = 2
    img, mask = (t.to(cuda) for t in make_batch(2, classes, 64, seed=31))
    img2, mask2 = (t.to(cuda) for t in make_batch(2, classes, 64, seed=32))
    keep = (torch.rand(2, 128, generator=torch.Generator().manual_seed(4)) < 0.8).float()
    if arch == 'deeplabv3plus':                      # element-wise pattern of ASPP.project's dropout, NHWC
        keep = (torch.rand(2, 4, 4, 256, generator=torch.Generator().manual_seed(4)) < 0.5).float()
    L.check(L.lib().octseg_set_deterministic(1))
    try:
        def run(graph):
            net = SegNet(arch, enc, classes=classes, device=cuda, compute_dtype=dtype, seed=11).train()
            net.use_train_graph = graph
            net.dropout_keep = keep
            if enc.startswith('efficientnet'):       # the id skips' drop_connect pattern: fixed, so that both runs see the same one (9 blocks of B0)
                net.drop_connect_keep = (torch.rand(9, 2, generator=torch.Generator().manual_seed(4)) < 0.9).float()
            out = []
            for k, (im, mk) in enumerate(((img, mask), (img, mask), (img2, mask2), (img, mask))):
                if k == 3:                              # parameters changed in place between replays: the graph repacks the weight images
                    with torch.no_grad():
                        net.arena.data.mul_(1.01)
                    net.params_changed()
                loss, logits, stats = net.train_step_raw(im, mk, normalize=True, mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225], grad_scale=0.5)
                torch.cuda.synchronize()
                out.append((loss.clone(), logits.clone(), stats.clone(), net.arena.grad.clone(), net.bn_buffers.clone()))
            return out
        eager, graph = run(False), run(True)
    finally:
        L.check(L.lib().octseg_set_deterministic(0))
    for k, (a, b) in enumerate(zip(eager, graph)):       # k = 0 eager warm-up, 1 capture + first launch, 2 and 3 replays
        for name, x, y in zip(('loss', 'logits', 'stats', 'grads', 'bn_buffers'), a, b):
            assert torch.equal(x, y), f'step {k}: {name} differs between the captured and the eager step'
    assert not torch.equal(eager[1][3], eager[2][3]) and not torch.equal(eager[1][3], eager[3][3])


@pytest.mark.parametrize('arch', ['FPN', 'DeepLabV3Plus', 'PSPNet', 'DeepLabV3'])
def test_sweep_architectures_through_the_model_class(cuda, tmp_path, arch):
    """The reference's sweep passes `architecture` straight to smp.create_model (configs/tune.yaml:9-18, model.py:38-44): the mirror class
    trains, validates, checkpoints and predicts with the two sweep architectures of round 3 exactly as with the BASELINE trio."""
    from oct_segmentation_amd.config import load_config
    from oct_segmentation_amd.predict import load_model
    from oct_segmentation_amd.train import fit
    cfg = load_config('train', [f'architecture={arch}', 'encoder=resnet18', 'epochs=2', 'input_size=64', 'batch_size=2', 'lr=0.001',
                                'compute_dtype=bf16', 'use_augmentation=false'])
    cfg['classes'] = ['Lipid core', 'Fibrous cap']
    batches = [tuple(t.to(cuda) for t in make_batch(2, 2, 64, seed=s)) for s in (1, 2, 3)]
    model, hist = fit(cfg, batches, val_batches=batches[:1], device=cuda, model_dir=str(tmp_path))
    assert len(hist) == 2 and all(np.isfinite(float(h['train']['loss'])) for h in hist)
    m2, _ = load_model(str(tmp_path), 'cuda', torch.float16)          # serving dtype
    out = m2.predict(np.zeros((1, 64, 64, 3), np.float32), 'cuda')
    assert out.shape == (1, 64, 64, 2) and set(np.unique(out)) <= {0.0, 1.0}


def test_eval_forward_after_a_replayed_training_step_through_the_c_abi(cuda):
    """a replay of octseg_net_train_step repacks the UNFOLDED training weight images inside the graph, behind the host's
    pack cache.  A C-ABI caller's sequence eval forward -> replayed training steps -> eval forward (no octseg_plan_params_changed in
    between; the Python wrapper always issues one) must not take a stale cache hit: the second eval forward has to equal the one after
    an explicit invalidation."""
    import ctypes as C
    from oct_segmentation_amd import _lib as L
    from oct_segmentation_amd.engine import SegNet
    net = SegNet('unet', 'resnet18', classes=1, device=cuda, compute_dtype=torch.bfloat16, seed=5)
    img, mask = (t.to(cuda) for t in make_batch(2, 1, 64, seed=3))
    plan = net._plan(2, 64, 64)
    lib, ws = L.lib(), plan.ws(cuda)
    m, s = (C.c_float * 3)(0.485, 0.456, 0.406), (C.c_float * 3)(0.229, 0.224, 0.225)
    logits = torch.empty(2, 1, 64, 64, device=cuda)
    loss = torch.empty((), device=cuda)
    stats = torch.empty(2, 1, 4, dtype=torch.int64, device=cuda)
    grads = torch.zeros_like(net.arena.data)
    st = torch.cuda.Stream(device=cuda)

    def eval_fwd():
        out = torch.empty_like(logits)
        with torch.cuda.stream(st):
            L.check(lib.octseg_net_forward(plan.handle, L.ptr(net.arena.data), L.ptr(net.bn_buffers), L.ptr(ws), L.ptr(img), L.ptr(out), 1, m, s, 0,
                                           L.stream_ptr()))
        st.synchronize()
        return out

    torch.cuda.synchronize()
    eval_fwd()                                                          # folded eval images cached (packed_fold = 1)
    L.check(lib.octseg_plan_set_train_graph(plan.handle, 1))
    for _ in range(4):                                                  # eager warm-up, capture + launch, two replays
        with torch.cuda.stream(st):
            L.check(lib.octseg_net_train_step(plan.handle, L.ptr(net.arena.data), L.ptr(grads), L.ptr(net.bn_buffers), L.ptr(ws), L.ptr(img),
                                              L.ptr(mask), L.ptr(logits), L.ptr(loss), L.ptr(stats), 1, m, s, 1.0, L.stream_ptr()))
        st.synchronize()
    a = eval_fwd()                                                      # no params_changed: must repack by itself
    L.check(lib.octseg_plan_params_changed(plan.handle))
    b = eval_fwd()
    assert torch.isfinite(a).all() and torch.equal(a, b)
