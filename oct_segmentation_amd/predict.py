"""Ensemble inference, host mirror of the reference's ``src/predict.py:23-101``.

Per class the reference loads ``<models_dir>/<LM|FC_LC|VV>/{config.json,weights.ckpt}``, resizes every image
to that model's ``input_size``, runs ``model.predict`` frame by frame (no normalisation, sigmoid > 0.5),
nearest-resizes the mask to ``output_size`` and writes channel ``MODELS_META[class]['index']`` into
``mask[:, :, CLASS_ID - 1]``.  Same semantics here; two deliberate host-side differences, both
result-neutral: the FC_LC network is run once for its two classes instead of twice,
and frames go through the engine in batches.  cv2 is absent in this image, so the two OpenCV resizes on the path
are restated from OpenCV 4.8.1 (``environment.yaml:23``) ``modules/imgproc/src/resize.cpp`` in integer / double
arithmetic: ``INTER_NEAREST`` of the predicted mask (``resizeNN``: ``min(floor(x * (1 / (dst / src))), src - 1)``) as
index tables for the GPU epilogue, and the default 8-bit ``INTER_LINEAR`` of ``preprocessing_img`` (``resizeGeneric_``
with 11-bit fixed-point coefficients, no antialiasing) in numpy.
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


def cv2_linear_coeffs(src, dst, horizontal=True):
    """Per output coordinate: (first tap index, second tap index, 11-bit coefficient pair) of OpenCV's 8-bit INTER_LINEAR
    (resize.cpp ``resize_``: ``fx = (float)((dx + 0.5) * scale - 0.5)``, ``saturate_cast<short>(coef * INTER_RESIZE_COEF_SCALE)``,
    INTER_RESIZE_COEF_SCALE = 2048).  Horizontal taps that leave the row get fx = 0; the vertical table keeps its fraction
    and clips the two row indices (resizeGeneric_Invoker)."""
    scale = 1.0 / (dst / float(src))                      # scale_x = 1. / inv_scale_x, both double
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)      # the (float) cast
    s0 = np.floor(f).astype(np.int64)
    f = (f - s0.astype(np.float32)).astype(np.float32)
    if horizontal:
        lo = s0 < 0
        f[lo] = 0.0; s0[lo] = 0
        hi = s0 >= src - 1
        f[hi] = 0.0; s0[hi] = src - 1
    s1 = np.clip(s0 + 1, 0, src - 1)
    s0 = np.clip(s0, 0, src - 1)
    one = np.float32(1.0)
    a0 = np.rint((one - f) * np.float32(2048.0)).astype(np.int64)   # cvRound: round half to even, as np.rint
    a1 = np.rint(f * np.float32(2048.0)).astype(np.int64)
    return s0, s1, a0, a1


def _is_exact_half(src, dst):
    """OpenCV's ``is_area_fast && iscale == 2`` per axis: scale = 1. / ((double)dst / src), iscale = saturate_cast<int>(scale) (= cvRound),
    |scale - iscale| < DBL_EPSILON."""
    scale = 1.0 / (dst / float(src))
    return int(np.rint(scale)) == 2 and abs(scale - 2.0) < np.finfo(np.float64).eps


def cv2_resize_linear_u8(img, dst_w, dst_h):
    """``cv2.resize(img_u8, (dst_w, dst_h))`` (default INTER_LINEAR) restated: horizontal pass in int32 with 11-bit
    coefficients, vertical pass ``(((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2`` (VResizeLinear<uchar>); an exact 2x
    decimation in both axes (e.g. 1024 -> 512) takes OpenCV's INTER_AREA fast path instead, as cv::resize does."""
    img = np.asarray(img)
    assert img.dtype == np.uint8
    squeeze = img.ndim == 2
    if squeeze:
        img = img[:, :, None]
    H, W = img.shape[:2]
    if _is_exact_half(W, dst_w) and _is_exact_half(H, dst_h):
        # cv::resize switches INTER_LINEAR to INTER_AREA for an exact 2x decimation (resize.cpp: `interpolation == INTER_LINEAR &&
        # is_area_fast && iscale_x == 2 && iscale_y == 2`); the 8-bit fast path (ResizeAreaFastVec, cn = 1 / 3 / 4) averages each
        # 2x2 block as (a + b + c + d + 2) >> 2
        s = img.astype(np.int64)
        out = ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)
        return out[:, :, 0] if squeeze else out
    x0, x1, ax0, ax1 = cv2_linear_coeffs(W, dst_w)
    y0, y1, ay0, ay1 = cv2_linear_coeffs(H, dst_h, horizontal=False)
    src = img.astype(np.int64)
    rows = src[:, x0, :] * ax0[None, :, None] + src[:, x1, :] * ax1[None, :, None]      # [H, dst_w, C], <= 255 * 2048
    S0, S1 = rows[y0], rows[y1]
    out = (((ay0[:, None, None] * (S0 >> 4)) >> 16) + ((ay1[:, None, None] * (S1 >> 4)) >> 16) + 2) >> 2
    out = np.clip(out, 0, 255).astype(np.uint8)
    return out[:, :, 0] if squeeze else out


def preprocessing_img(img, input_size):
    """data/utils.py:159-166: np.array(img) -> cvtColor RGB2BGR -> cv2.resize(image, (input_size, input_size))."""
    image = np.asarray(img.convert('RGB'))[:, :, ::-1]
    return cv2_resize_linear_u8(np.ascontiguousarray(image), input_size, input_size)


def cv2_nearest_index(src, dst):
    """Source index of every output coordinate of ``cv2.resize(..., interpolation=cv2.INTER_NEAREST)`` (predict.py:92-96).
    OpenCV's ``resizeNN``: ``ifx = 1. / fx`` with ``fx = (double)dst / src``, ``sx = min(cvFloor(x * ifx), src - 1)`` --
    no half-pixel centre (INTER_NEAREST_EXACT would have one; the reference does not use it)."""
    ifx = 1.0 / (float(dst) / float(src))
    x = np.arange(dst, dtype=np.float64)
    return np.minimum(np.floor(x * ifx).astype(np.int64), src - 1).astype(np.int32)


def pil_nearest_index(src, dst):
    """Source index of every output pixel of ``PIL.Image.resize((dst, ...), NEAREST)`` (Pillow steps a double by src/dst from
    src/dst/2 and truncates).  NOT the rule of ``segment()`` -- kept for callers that resample label images with Pillow."""
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

    Every model runs once (the reference runs FC_LC once per class), in batches (with ``use_graph`` the nets' replayed forwards side by
    side); thresholding, the nearest resize to
    ``output_size`` and the 4-channel mask assembly happen on the GPU (``octseg_mask_assemble``); one D2H copy of the
    assembled 0/1 stack at the end instead of one logits tensor per frame and class."""
    from . import _lib as L
    n = len(images)
    # cv2 sizes are (width, height); the reference allocates masks as [output_size[0], output_size[1], 4] and resizes
    # to tuple(output_size): the assignment into mask[:, :, c] only works for square sizes, and so do the extents below
    oh, ow = masks[0].shape[0], masks[0].shape[1]
    stack = torch.zeros((n, oh, ow, 4), dtype=torch.float32, device=device)
    cache, tables, loaded = {}, {}, {}
    for class_name in classes:     # every distinct model once: weights and the preprocessed frames at that model's input size
        model_dir = os.path.join(models_dir, MODELS_META[class_name]['model_dir'])
        if model_dir not in loaded:
            model, cfg = load_model(model_dir, device, compute_dtype, use_graph=use_graph)
            loaded[model_dir] = (model, np.array([preprocessing_img(img, cfg['input_size']) for img in images]))
    parts = {d: [] for d in loaded}
    for i in range(0, n, batch_size):
        if use_graph:
            # the replayed forwards of the (up to three) nets are started together, each on its plan's own stream, and joined in order:
            # at one frame per step a single net fills a fraction of the chip (SegNet.forward_async)
            handles = {}
            for d, (model, batch) in loaded.items():
                x = torch.as_tensor(np.ascontiguousarray(batch[i:i + batch_size].transpose((0, 3, 1, 2))), dtype=torch.float32).to(model.model.device)
                handles[d] = model.model.eval().forward_async(x, normalize=False)
            for d, h in handles.items():
                parts[d].append(loaded[d][0].model.forward_join(h))
        else:
            for d, (model, batch) in loaded.items():
                parts[d].append(model.predict_logits(batch[i:i + batch_size]))
    for d in loaded:
        cache[d] = torch.cat(parts[d], dim=0)
    del loaded
    for class_name in classes:
        meta = MODELS_META[class_name]
        model_dir = os.path.join(models_dir, meta['model_dir'])
        z = cache[model_dir]
        ch = meta['index'] if z.shape[1] > 1 else 0
        key = (z.shape[2], z.shape[3])
        if key not in tables:   # cv2.resize(predict_mask, tuple(output_size), INTER_NEAREST): resizeNN's row / column tables
            tables[key] = (torch.from_numpy(cv2_nearest_index(z.shape[2], oh)).to(device),
                           torch.from_numpy(cv2_nearest_index(z.shape[3], ow)).to(device))
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
