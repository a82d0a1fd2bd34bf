"""``OCTSegmentationModel`` -- host-side mirror of the reference task module
(``src/models/smp/model.py:18-200``) over the gfx950 engine.

Same constructor keywords, same methods (``forward``, ``training_step``,
``validation_step``, ``configure_optimizers``, ``predict``,
``load_from_checkpoint``), same ``state_dict`` keys (``model.*``, ``mean``,
``std``).  Lightning, W&B, cv2 and the per-epoch image dump are not part of the
hot path and are not reproduced (out of scope).
"""
import pickle

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from .engine import SegNet, get_preprocessing_params
from .metrics import DeferredMetrics, get_metrics_from_stats

# reference src/data/utils.py:16-45
CLASS_IDS = {'Lumen': 1, 'Fibrous cap': 2, 'Lipid core': 3, 'Vasa vasorum': 4}


class DiceLoss:
    """Marker object kept for API parity (``self.loss_fn``); the loss itself is fused into the
    engine (``SegNet.dice_step``): smp DiceLoss(MULTILABEL_MODE, from_logits=True), model.py:55.
    ``kind``: 'dice' (the reference) | 'bce' (F.binary_cross_entropy_with_logits, mean) | 'dice+bce' (their sum) --
    evaluated and differentiated by the same fused pass (octseg_plan_set_loss)."""
    mode = 'multilabel'
    from_logits = True
    smooth = 0.0
    eps = 1e-7

    def __init__(self, kind='dice'):
        self.kind = kind


class _LiveRangeOptimizer:
    """torch optimizer over the whole arena of a net with never-run parameters (PSPNet keeps encoder.layer3 / layer4 in its
    state_dict, smp encoder_depth 3).  In the reference those parameters have ``grad is None`` and every torch optimizer skips
    them; here the arena is ONE parameter whose gradient is zero there, so weight decay (and Adam's moments) would still move
    them: the dead ranges are put back after every step."""

    def __init__(self, opt, net):
        self._opt, self._net = opt, net
        lo, self._dead = 0, []
        for a, b in net.live_ranges:
            if a > lo:
                self._dead.append((lo, a))
            lo = b
        if lo < net.param_numel:
            self._dead.append((lo, net.param_numel))

    def step(self, *args, **kwargs):
        keep = [self._net.arena.data[a:b].clone() for a, b in self._dead]
        out = self._opt.step(*args, **kwargs)
        for (a, b), k in zip(self._dead, keep):
            self._net.arena.data[a:b] = k
        return out

    def __getattr__(self, name):
        return getattr(self._opt, name)


class FusedOptimizer:
    """torch.optim-like wrapper of ``octseg_optim_step`` over the flat parameter arena
    (SGD | Adam | RMSprop | RAdam with torch defaults, coupled L2 weight decay; model.py:150-181)."""

    def __init__(self, net, name, lr, weight_decay):
        if name not in L.OPT_KINDS:
            raise ValueError(f'Unknown optimizer: {name}')
        self.net, self.name, self.kind = net, name, L.OPT_KINDS[name]
        self.lr, self.weight_decay = float(lr), float(weight_decay)
        self.step_count = 0
        n, dev = net.param_numel, net.device
        self.m = torch.zeros(n, dtype=torch.float32, device=dev) if self.kind in (1, 3) else None
        self.v = torch.zeros(n, dtype=torch.float32, device=dev) if self.kind in (1, 2, 3) else None
        self.param_groups = [{'params': [net.arena], 'lr': self.lr, 'weight_decay': self.weight_decay}]

    def zero_grad(self, set_to_none=True):
        self.net.arena.grad = None

    def step(self, grad_scale=1.0):
        g = self.net.arena.grad
        if g is None:
            raise RuntimeError('optimizer.step() called before any backward')
        self.step_count += 1
        self.lr = float(self.param_groups[0]['lr'])
        for lo, hi in getattr(self.net, 'live_ranges', [(0, self.net.param_numel)]):     # (one range for every net but PSPNet: engine.py)
            L.check(L.lib().octseg_optim_step(self.kind, L.ptr(self.net.arena.data[lo:hi]), L.ptr(g[lo:hi]),
                                              L.ptr(None if self.m is None else self.m[lo:hi]), L.ptr(None if self.v is None else self.v[lo:hi]),
                                              hi - lo, self.lr, self.weight_decay, self.step_count, float(grad_scale), L.stream_ptr()))
        self.net.params_changed()

    def state_dict(self):
        return {'name': self.name, 'step': self.step_count, 'lr': self.lr, 'weight_decay': self.weight_decay,
                'm': self.m, 'v': self.v}


class OCTSegmentationModel(nn.Module):
    """The model dedicated to the segmentation of OCT images (MI355X engine)."""

    def __init__(self, arch, encoder_name, model_name, in_channels, classes, lr=0.0001, data_dir=None,
                 weight_decay=0.0001, optimizer_name='Adam', input_size=512, img_save_interval=1,
                 save_wandb_media=False, device='cuda', compute_dtype=torch.bfloat16, fused_optimizer=True, encoder_weights=None,
                 defer_metrics=False, loss='dice', **kwargs):
        super().__init__()
        # **kwargs go to the network factory as in the reference (model.py:38-44 -> smp.create_model); SegNet rejects what it
        # does not implement.  encoder_weights: see SegNet (the reference's smp default 'imagenet' needs a download).
        self.model = SegNet(arch, encoder_name, encoder_weights=encoder_weights, in_channels=in_channels, classes=len(classes),
                            device=device, compute_dtype=compute_dtype, loss=loss, **kwargs)
        self.classes = list(classes)
        self.data_dir = data_dir
        self.epoch = 0
        params = get_preprocessing_params(encoder_name)
        dev = self.model.device
        self.register_buffer('std', torch.tensor(params['std'], device=dev).view(1, 3, 1, 1))
        self.register_buffer('mean', torch.tensor(params['mean'], device=dev).view(1, 3, 1, 1))
        self._mean, self._std = list(params['mean']), list(params['std'])
        self.training_step_outputs = []
        self.validation_step_outputs = []
        # defer_metrics=False keeps the reference's per-step dict (one D2H copy of the counts per step, utils.py:25-35);
        # True keeps the counts and the loss on the GPU until flush_metrics() -- one copy per epoch, same rows
        self.defer_metrics = bool(defer_metrics)
        self._deferred = {'train': DeferredMetrics(), 'test': DeferredMetrics()}
        self.validation_best_metrics = {}
        self.loss_fn = DiceLoss(loss)
        self.model_name = model_name
        self.lr, self.weight_decay, self.optimizer = lr, weight_decay, optimizer_name
        self.input_size = input_size
        self.img_save_interval, self.save_wandb_media = img_save_interval, save_wandb_media
        self.fused_optimizer = fused_optimizer
        self.class_values = [CLASS_IDS[cl] for cl in self.classes if cl in CLASS_IDS]

    # ---- model.py:65-71: (image - mean) / std is fused into the stem's im2col load
    def forward(self, image):
        return self.model(image, normalize=True, mean=self._mean, std=self._std)

    def _step(self, batch):
        img, mask = batch
        loss, logits, stats = self.model.dice_step(img, mask, normalize=True, mean=self._mean, std=self._std)
        return loss, logits, stats

    # ---- model.py:73-95.  The thresholded mask and tp/fp/fn/tn come out of the Dice kernel.
    def record_step(self, split, stats, loss):
        """What get_metrics + the append in training_step / validation_step do (model.py:82-92, 116-126)."""
        if self.defer_metrics:
            self._deferred[split].append(stats, loss)
        else:
            (self.training_step_outputs if split == 'train' else self.validation_step_outputs).append(get_metrics_from_stats(stats, loss))

    def flush_metrics(self, split):
        """The split's per-step metric dicts of the epoch so far (deferred mode: the one host copy happens here)."""
        outputs = self.training_step_outputs if split == 'train' else self.validation_step_outputs
        if self.defer_metrics:
            outputs.extend(self._deferred[split].flush())
        return outputs

    def training_step(self, batch, batch_idx=0):
        loss, logits, stats = self._step(batch)
        self.record_step('train', stats, loss)
        return {'loss': loss}

    # ---- model.py:108-132
    def validation_step(self, batch, batch_idx=0):
        with torch.no_grad():
            loss, logits, stats = self._step(batch)
        self.record_step('test', stats, loss)
        if self.defer_metrics:   # no host copy: the batch F1 as a device scalar (2tp / (2tp + fp + fn), 0/0 -> 1e-7 as utils.py:17)
            tp, fp, fn = stats[..., 0].float(), stats[..., 1].float(), stats[..., 2].float()
            f1 = torch.nan_to_num(2 * tp / (2 * tp + fp + fn), nan=1e-7)
            return {'val/loss': loss, 'val/f1': f1.mean()}
        return {'val/loss': loss, 'val/f1': float(np.mean(self.validation_step_outputs[-1]['f1']).mean())}

    # ---- model.py:150-181
    def configure_optimizers(self):
        if self.optimizer not in ('SGD', 'RMSprop', 'RAdam', 'SAdam', 'Adam'):
            raise ValueError(f'Unknown optimizer: {self.optimizer}')
        if self.optimizer == 'SAdam':  # torch.optim.SparseAdam raises on dense grads in the reference too
            raise ValueError('SAdam (SparseAdam) cannot optimise dense gradients')
        if self.fused_optimizer:
            return FusedOptimizer(self.model, self.optimizer, self.lr, self.weight_decay)
        cls = {'SGD': torch.optim.SGD, 'RMSprop': torch.optim.RMSprop, 'RAdam': torch.optim.RAdam,
               'Adam': torch.optim.Adam}[self.optimizer]
        opt = cls(self.parameters(), lr=self.lr, weight_decay=self.weight_decay)
        if len(self.model.live_ranges) != 1 or self.model.live_ranges[0] != (0, self.model.param_numel):
            return _LiveRangeOptimizer(opt, self.model)
        return opt

    # ---- model.py:183-200: NHWC numpy in, no normalisation, sigmoid > 0.5, NHWC numpy out.  The threshold and the
    # NCHW -> NHWC transpose run in the engine's serving epilogue (octseg_mask_assemble at identity size): one D2H copy of
    # the 0/1 masks, no torch arithmetic.
    def predict(self, images, device='cuda'):
        z = self.predict_logits(images)
        n, c, h, w = z.shape
        out = torch.empty((n, h, w, c), dtype=torch.float32, device=z.device)
        for ch in range(c):
            L.check(L.lib().octseg_mask_assemble(L.ptr(z), n, c, h, w, ch, L.ptr(out), h, w, c, ch, None, None, L.stream_ptr()))
        return out.cpu().numpy().round()

    def predict_logits(self, images):
        """The device half of ``predict``: NHWC numpy in, NCHW float32 logits on the GPU out (no host round trip)."""
        x = torch.as_tensor(np.ascontiguousarray(images.transpose((0, 3, 1, 2))), dtype=torch.float32).to(self.model.device)
        was_training = self.model.training
        try:
            return self.model(x, normalize=False)
        finally:
            self.model.train(was_training)

    @staticmethod
    def to_tensor_shape(x):
        return x.transpose([2, 0, 1]).astype('float32')

    # ---- checkpoint compatibility (predict.py:39-48): Lightning .ckpt = pickled dict with 'state_dict'
    def state_dict(self, *args, **kwargs):
        sd = self.model.state_dict(prefix='model.')
        sd['std'] = self.std.clone()
        sd['mean'] = self.mean.clone()
        return sd

    def load_state_dict(self, state_dict, strict=True):
        inner = {k[len('model.'):]: v for k, v in state_dict.items() if k.startswith('model.')}
        res = self.model.load_state_dict(inner, strict=strict)
        for k in ('mean', 'std'):
            if k in state_dict:
                getattr(self, k).copy_(state_dict[k].to(getattr(self, k).device))
        self._mean = [float(v) for v in self.mean.flatten().cpu()]
        self._std = [float(v) for v in self.std.flatten().cpu()]
        return res

    def save_checkpoint(self, path, epoch=0, global_step=0):
        ckpt = {'epoch': epoch, 'global_step': global_step, 'pytorch-lightning_version': '2.2.1',
                'state_dict': {k: v.cpu() for k, v in self.state_dict().items()}}
        torch.save(ckpt, path)

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, map_location=None, **kwargs):
        kwargs['encoder_weights'] = None   # predict.py:41 passes None: the checkpoint holds every weight
        if map_location not in (None, 'cpu'):
            kwargs.setdefault('device', map_location)
        try:
            ckpt = torch.load(checkpoint_path, map_location='cpu', weights_only=False)
        except pickle.UnpicklingError as e:  # pragma: no cover
            raise RuntimeError(f'cannot read checkpoint {checkpoint_path}: {e}')
        model = cls(**kwargs)
        model.load_state_dict(ckpt['state_dict'], strict=True)
        return model
