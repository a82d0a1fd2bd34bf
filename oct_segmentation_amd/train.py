"""Thin training loop standing in for ``pl.Trainer.fit`` in ``src/models/smp/train.py:25-134`` (Lightning,
W&B and the file dataset are out of scope): builds ``OCTSegmentationModel`` from a train.yaml-style config,
steps it on batches from any iterable of ``(img [B,3,S,S] f32 0..255 BGR, mask [B,C,S,S] f32)`` and writes the
reference's ``config.json`` + ``weights.ckpt`` pair.  Data parallel when launched under torchrun."""
import json
import os
import warnings

import numpy as np
import torch

from . import augment as augment_mod
from .metrics import aggregate_epoch, save_metrics_on_epoch
from .model import OCTSegmentationModel
from . import parallel


def write_model_config(cfg, model_dir):
    """train.py:105-119."""
    os.makedirs(model_dir, exist_ok=True)
    with open(os.path.join(model_dir, 'config.json'), 'w') as f:
        json.dump({'model_name': f"{cfg['architecture']}_{cfg['encoder']}", 'architecture': cfg['architecture'],
                   'encoder': cfg['encoder'], 'input_size': cfg['input_size'], 'classes': list(cfg['classes']),
                   'batch_size': cfg['batch_size'], 'optimizer': cfg['optimizer'], 'lr': cfg['lr']}, f, indent=2)


def write_epoch_rows(model_dir, classes, epoch, train_outputs, val_outputs, best_metrics):
    """metrics.csv rows of one epoch in the reference's order: Lightning runs the validation loop inside the training epoch, so
    ``on_validation_epoch_end`` (model.py:134-148: the 'test' rows + best metrics) writes BEFORE ``on_train_epoch_end``
    (model.py:97-106: the 'train' rows) -- eval/training/Lumen/fold_1/metrics.csv:2-3.  Returns the updated best metrics."""
    if val_outputs:
        _, best_metrics = save_metrics_on_epoch(val_outputs, 'test', model_dir, classes, epoch, best_metrics)
    save_metrics_on_epoch(train_outputs, 'train', model_dir, classes, epoch)
    return best_metrics


_warned_random_init = False


def fit(cfg, train_batches, val_batches=None, device='cuda', model_dir=None, augment_seed=None):
    """``cfg`` keys beyond train.yaml's: ``compute_dtype`` ('bf16' | 'fp32'), ``allreduce_slices``, ``encoder_weights`` (passed to
    the network factory: None = random init, a path / 'imagenet' = torchvision weights, see SegNet.load_encoder_weights) and
    ``defer_metrics`` (True: tp/fp/fn/tn and the loss of every step stay on the GPU and cross to the host ONCE per epoch instead
    of once per step -- the per-step ``.cpu()`` of utils.py:25-35 is a device sync; the epoch rows are identical)."""
    global _warned_random_init
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world > 1:
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError('WORLD_SIZE > 1: initialise torch.distributed (backend "nccl" = RCCL) and set the CUDA device of this '
                               'rank before fit(); every rank passes ITS shard of the batches (parallel.shard_range)')
        if dist.get_world_size() != world:
            raise RuntimeError(f'WORLD_SIZE={world} but the process group has {dist.get_world_size()} ranks')
    dt = torch.float32 if str(cfg.get('compute_dtype', 'bf16')) in ('fp32', 'float32') else torch.bfloat16
    enc_w = cfg.get('encoder_weights')
    if enc_w is None and not os.environ.get('OCTSEG_IMAGENET_DIR') and not _warned_random_init:
        # the reference's smp.create_model default is encoder_weights='imagenet' (model.py:38-44): say that this run differs
        warnings.warn("fit(): encoder_weights is None and OCTSEG_IMAGENET_DIR is unset -- the encoder starts from random init, "
                      "where the reference downloads ImageNet weights; pass cfg['encoder_weights']", stacklevel=2)
        _warned_random_init = True
    model = OCTSegmentationModel(cfg['architecture'], cfg['encoder'], f"{cfg['architecture']}_{cfg['encoder']}", 3,
                                 cfg['classes'], lr=cfg['lr'], weight_decay=cfg['weight_decay'],
                                 optimizer_name=cfg['optimizer'], input_size=cfg['input_size'], device=device, compute_dtype=dt,
                                 encoder_weights=enc_w, defer_metrics=bool(cfg.get('defer_metrics', False)))
    net = model.model
    exchange = None
    if world > 1:
        parallel.broadcast_parameters(net)
        # cfg['allreduce_dtype'] = 'bf16': gradients cross xGMI as bfloat16 (half the bytes of the per-link-bound ring)
        exchange = parallel.GradientExchange(net, nslices=int(cfg.get('allreduce_slices', 3)), wire_dtype=str(cfg.get('allreduce_dtype', 'fp32')))
    opt = model.configure_optimizers()
    history = []
    aug_rng = None
    rank0 = int(os.environ.get('RANK', '0')) == 0
    best_val = None   # ModelCheckpoint(monitor='val/loss', mode='min', save_top_k=1, filename='weights'), train.py:67-76
    if model_dir is not None and rank0:
        write_model_config(cfg, model_dir)   # the reference writes config.json before trainer.fit (train.py:105-119)
    for epoch in range(1, int(cfg['epochs']) + 1):
        model.train()
        model.training_step_outputs.clear()
        n_train = 0
        for img, mask in train_batches:
            n_train += 1
            if cfg.get('use_augmentation', False):   # train.yaml use_augmentation (dataset.py:119-123), on the GPU
                if aug_rng is None:
                    aug_rng = np.random.default_rng(augment_seed)
                img, mask = augment_mod.augment(img, mask, augment_mod.sample_params(img.shape[0], img.shape[-1], aug_rng))
            if world > 1:
                parallel.broadcast_buffers(net)
            # grad_scale = 1/W inside the backward + SUM all-reduce = DDP's gradient mean; the all-reduce runs slice by
            # slice beside the backward (parallel.GradientExchange)
            try:
                loss, logits, stats = net.train_step_raw(img, mask, normalize=True, mean=model._mean, std=model._std,
                                                         grad_scale=1.0 / world, exchange=exchange)
            except parallel.GradientExchangeError as e:   # the group is aborted: only a fresh process is a clean state
                parallel.exit_on_exchange_failure(e)
            opt.step()
            model.record_step('train', stats, loss)
        if n_train == 0:
            raise ValueError(f'fit(): train_batches yielded no batches in epoch {epoch}; pass a re-iterable (a list or a DataLoader), '
                             f'not a generator that is exhausted after the first epoch')
        row = {'epoch': epoch}
        val_outputs = None
        if val_batches is not None:   # Lightning runs the validation loop before on_train_epoch_end
            model.eval()
            model.validation_step_outputs.clear()
            vloss, vcount = [], []
            for batch in val_batches:
                out = model.validation_step(batch)
                vloss.append(out['val/loss'].detach().reshape(1))   # self.log('val/loss', on_epoch=True): batch-size-weighted mean
                vcount.append(batch[0].shape[0])
            if not vloss:
                raise ValueError(f'fit(): val_batches yielded no batches in epoch {epoch}; pass a re-iterable (a list or a DataLoader), '
                                 f'not a generator that is exhausted after the first epoch')
            w = torch.tensor(vcount, dtype=torch.float64)
            row['val/loss'] = float((torch.cat(vloss).double().cpu() * w).sum() / max(float(w.sum()), 1.0))   # one D2H per epoch
            if model_dir is not None and rank0 and (best_val is None or row['val/loss'] < best_val):
                best_val = row['val/loss']      # only the best-validation-loss epoch is kept, as predict.py then loads it
                model.save_checkpoint(os.path.join(model_dir, 'weights.ckpt'), epoch=epoch)
            val_outputs = model.flush_metrics('test')
            row['test'] = aggregate_epoch(val_outputs)   # the reference calls the split 'test'
        train_outputs = model.flush_metrics('train')
        row['train'] = aggregate_epoch(train_outputs)
        if model_dir is not None and rank0:
            model.validation_best_metrics = write_epoch_rows(model_dir, cfg['classes'], epoch, train_outputs, val_outputs,
                                                             model.validation_best_metrics)
        history.append(row)
    if model_dir is not None and rank0 and val_batches is None:   # nothing to monitor: keep the last epoch
        model.save_checkpoint(os.path.join(model_dir, 'weights.ckpt'), epoch=len(history))
    return model, history
