"""Ensemble inference, host mirror of the reference's ``src/predict.py:23-101``.

Per class the reference loads ``<models_dir>/<LM|FC_LC|VV>/{config.json,weights.ckpt}``, resizes every image
to that model's ``input_size``, runs ``model.predict`` frame by frame (no normalisation, sigmoid > 0.5),
nearest-resizes the mask to ``output_size`` and writes channel ``MODELS_META[class]['index']`` into
``mask[:, :, CLASS_ID - 1]``.  Same semantics here; two deliberate host-side differences, both
result-neutral: the FC_LC network is run once for its two classes instead of twice,
and frames go through the engine in batches.  cv2 is absent in this image: resizing uses PIL (bilinear /
nearest), so preprocessing agrees with the reference statistically, not bit for bit.
"""
import json
import os

import numpy as np
import torch
from PIL import Image

from .model import CLASS_IDS, OCTSegmentationModel

MODELS_META = {
    'Lumen': {'model_dir': 'LM', 'index': 0},
    'Lipid core': {'model_dir': 'FC_LC', 'index': 0},
    'Fibrous cap': {'model_dir': 'FC_LC', 'index': 1},
    'Vasa vasorum': {'model_dir': 'VV', 'index': 0},
}


def load_model(model_dir, device='cuda', compute_dtype=torch.bfloat16, use_graph=False):
    """predict.py:31-50."""
    with open(os.path.join(model_dir, 'config.json')) as f:
        cfg = json.load(f)
    model = OCTSegmentationModel.load_from_checkpoint(
        checkpoint_path=os.path.join(model_dir, 'weights.ckpt'), encoder_weights=None, arch=cfg['architecture'],
        encoder_name=cfg['encoder'], model_name=cfg['model_name'], in_channels=3, classes=cfg['classes'],
        map_location=device, compute_dtype=compute_dtype)
    model.eval()
    # serving option: every eval forward of a (B, H, W) plan replays one captured hipGraph (bit-exact; measured no faster
    # than the eager launches on MI355X -- 5.87 vs 5.74 ms at B=1 704x704 -- so it is off unless asked for)
    model.model.use_graph = bool(use_graph)
    return model, cfg


def preprocessing_img(img, input_size):
    """data/utils.py:159-166: RGB -> BGR, resize to input_size (bilinear)."""
    img = img.convert('RGB').resize((input_size, input_size), Image.BILINEAR)
    return np.asarray(img)[:, :, ::-1].copy()


def pil_nearest_index(src, dst):
    """Source index of every output pixel of ``PIL.Image.resize((dst, ...), NEAREST)``: Pillow steps a double by
    src/dst from src/dst/2 and truncates, so exact-integer positions may fall on either side; same arithmetic here."""
    a0 = src / dst
    xx = a0 * 0.5
    out = np.empty(dst, dtype=np.int32)
    for i in range(dst):
        out[i] = min(int(xx), src - 1)
        xx += a0
    return out


def segment(images, masks, output_size, classes, models_dir, device='cuda', batch_size=8, compute_dtype=torch.bfloat16,
            use_graph=False):
    """predict.py:61-101.  images: list of PIL images; masks: list of zero arrays [H_out, W_out, 4].

    Every model runs once (the reference runs FC_LC once per class), in batches; thresholding, the nearest resize to
    ``output_size`` and the 4-channel mask assembly happen on the GPU (``octseg_mask_assemble``); one D2H copy of the
    assembled 0/1 stack at the end instead of one logits tensor per frame and class."""
    from . import _lib as L
    n = len(images)
    # PIL sizes are (width, height); the reference allocates masks as [output_size[0], output_size[1], 4] and resizes
    # to tuple(output_size): it only ever uses square sizes, and so do the extents below
    oh, ow = masks[0].shape[0], masks[0].shape[1]
    stack = torch.zeros((n, oh, ow, 4), dtype=torch.float32, device=device)
    cache, tables = {}, {}
    for class_name in classes:
        meta = MODELS_META[class_name]
        model_dir = os.path.join(models_dir, meta['model_dir'])
        if model_dir not in cache:
            model, cfg = load_model(model_dir, device, compute_dtype, use_graph=use_graph)
            batch = np.array([preprocessing_img(img, cfg['input_size']) for img in images])
            logits = [model.predict_logits(batch[i:i + batch_size]) for i in range(0, n, batch_size)]
            cache[model_dir] = torch.cat(logits, dim=0)
            del model
        z = cache[model_dir]
        ch = meta['index'] if z.shape[1] > 1 else 0
        key = (z.shape[2], z.shape[3])
        if key not in tables:   # PIL size = (width, height) = tuple(output_size): columns follow output_size[0]
            tables[key] = (torch.from_numpy(pil_nearest_index(z.shape[2], oh)).to(device),
                           torch.from_numpy(pil_nearest_index(z.shape[3], ow)).to(device))
        rows, cols = tables[key]
        L.check(L.lib().octseg_mask_assemble(L.ptr(z), n, z.shape[1], z.shape[2], z.shape[3], int(ch), L.ptr(stack), oh, ow, 4,
                                             CLASS_IDS[class_name] - 1, L.ptr(rows), L.ptr(cols), L.stream_ptr()))
    host = stack.cpu().numpy()
    for i, mask in enumerate(masks):
        for class_name in classes:
            c = CLASS_IDS[class_name] - 1
            mask[:, :, c] = host[i, :, :, c]
    return masks


def data_processing(image_paths, output_size):
    """data/utils.py:169-192 without the directory creation."""
    images, masks = [], []
    for p in image_paths:
        images.append(Image.open(p).resize(tuple(output_size)))
        masks.append(np.zeros((output_size[0], output_size[1], 4)))
    return images, masks
