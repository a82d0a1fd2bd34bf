"""Training augmentation on the GPU -- host half.

Mirror of ``OCTDataset.get_img_augmentation`` (reference ``src/models/smp/dataset.py:160-207``): the same eight
transforms, probabilities and parameter ranges, drawn per frame on the host; the geometric ones are composed into one
inverse homography and everything is applied by ONE kernel (``octseg_augment``, ``csrc/augment.hip``).  The reference
resamples the uint8 image once per geometric transform; one interpolation here -- statistical, not bit, parity.

    params = sample_params(B, S, rng)            # numpy [B, NPARAM] float32
    img2, mask2 = augment(img, mask, params)     # CUDA tensors [B,3,S,S] / [B,C,S,S] float32
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L

NPARAM = 36

# reference dataset.py:166-205
P_HFLIP = 0.50
P_SSR, SHIFT_LIMIT, SCALE_LIMIT, ROTATE_LIMIT = 0.20, 0.0625, 0.1, 15.0
P_CROP, CROP_MIN, CROP_MAX = 0.20, 0.8, 0.9
P_NOISE, NOISE_VAR = 0.15, (1.5, 6.5)
P_PERSPECTIVE, PERSPECTIVE_SCALE = 0.20, (0.05, 0.1)
P_BC, BRIGHTNESS_LIMIT, CONTRAST_LIMIT = 0.15, 0.15, 0.15
P_HSV, HUE_LIMIT, SAT_LIMIT, VAL_LIMIT = 0.15, 15.0, 20.0, 15.0


def _translate(tx, ty):
    return np.array([[1, 0, tx], [0, 1, ty], [0, 0, 1]], dtype=np.float64)


def _homography_from_points(src, dst):
    """3x3 H with H @ [x, y, 1] ~ [u, v, 1] for four point pairs (what cv2.getPerspectiveTransform solves)."""
    A, b = [], []
    for (x, y), (u, v) in zip(src, dst):
        A.append([x, y, 1, 0, 0, 0, -u * x, -u * y]); b.append(u)
        A.append([0, 0, 0, x, y, 1, -v * x, -v * y]); b.append(v)
    h = np.linalg.solve(np.array(A, dtype=np.float64), np.array(b, dtype=np.float64))
    return np.append(h, 1.0).reshape(3, 3)


def sample_frame(S, rng, crop_frac=None):
    """One frame's decisions -> (forward homography source->output, post-crop homography, crop window, alpha, beta, sigma,
    seed, hue, sat, val, hsv_on, log).

    ``crop_frac``: the reference draws the RandomCrop size ONCE per transform construction with Python's ``random``
    (dataset.py:175-179) -- i.e. per sample, since the Compose is rebuilt in every ``__getitem__``."""
    log = {}
    M = np.eye(3)
    post = np.eye(3)                                  # transforms applied AFTER crop + pad (Perspective)
    rect = (-1e9, -1e9, 1e9, 1e9)
    c = S / 2 - 0.5                                   # albumentations' centre for rotation / scaling
    if rng.random() < P_HFLIP:                        # x -> S - 1 - x
        M = np.array([[-1, 0, S - 1], [0, 1, 0], [0, 0, 1]], dtype=np.float64) @ M
        log['hflip'] = True
    if rng.random() < P_SSR:
        angle = rng.uniform(-ROTATE_LIMIT, ROTATE_LIMIT)
        scale = 1.0 + rng.uniform(-SCALE_LIMIT, SCALE_LIMIT)
        dx, dy = rng.uniform(-SHIFT_LIMIT, SHIFT_LIMIT) * S, rng.uniform(-SHIFT_LIMIT, SHIFT_LIMIT) * S
        a = np.deg2rad(angle)
        # cv2.getRotationMatrix2D(centre, angle, scale): counter-clockwise for positive angles in image coordinates
        R = np.array([[scale * np.cos(a), scale * np.sin(a), 0], [-scale * np.sin(a), scale * np.cos(a), 0], [0, 0, 1]])
        M = _translate(dx, dy) @ _translate(c, c) @ R @ _translate(-c, -c) @ M
        log['ssr'] = (angle, scale, dx, dy)
    if rng.random() < P_CROP:
        f_h, f_w = crop_frac if crop_frac is not None else (rng.uniform(CROP_MIN, CROP_MAX), rng.uniform(CROP_MIN, CROP_MAX))
        ch, cw = int(f_h * S), int(f_w * S)
        y0 = int(rng.random() * (S - ch + 1)); x0 = int(rng.random() * (S - cw + 1))
        pad_t, pad_l = (S - ch) // 2, (S - cw) // 2   # PadIfNeeded: centred, constant 0
        # crop then pad = a shift by (pad - origin) with everything outside the crop window blank: the blanking is what the
        # kernel's constant-0 border does NOT give for free, so the crop is expressed as a shift and the window as a mask
        M = _translate(pad_l - x0, pad_t - y0) @ M
        rect = (pad_l, pad_t, pad_l + cw, pad_t + ch)
        log['crop'] = (y0, x0, ch, cw, pad_t, pad_l)
    if rng.random() < P_PERSPECTIVE:
        s = rng.uniform(*PERSPECTIVE_SCALE)
        # albumentations / imgaug: every corner moves inwards by |N(0, s)| (clipped) of the frame size
        jit = np.clip(np.abs(rng.normal(0.0, s, size=(4, 2))), 0.0, 0.49) * S
        src = np.array([[0, 0], [S - 1, 0], [S - 1, S - 1], [0, S - 1]], dtype=np.float64)
        moved = src + jit * np.array([[1, 1], [-1, 1], [-1, -1], [1, -1]])
        P = _homography_from_points(moved, src)       # fit_output=False: the jittered quad is stretched back to the frame
        M = P @ M
        post = P @ post
        log['perspective'] = s
        log['perspective_matrix'] = P
    alpha, beta = 1.0, 0.0
    sigma, seed = 0.0, 0
    if rng.random() < P_NOISE:
        sigma = float(np.sqrt(rng.uniform(*NOISE_VAR)))
        seed = int(rng.integers(1, 2 ** 31 - 1))
        log['noise'] = sigma
    if rng.random() < P_BC:
        alpha = 1.0 + rng.uniform(-CONTRAST_LIMIT, CONTRAST_LIMIT)
        beta = rng.uniform(-BRIGHTNESS_LIMIT, BRIGHTNESS_LIMIT)
        log['bc'] = (alpha, beta)
    hue = sat = val = 0.0
    hsv_on = False
    if rng.random() < P_HSV:
        hue, sat, val = rng.uniform(-HUE_LIMIT, HUE_LIMIT), rng.uniform(-SAT_LIMIT, SAT_LIMIT), rng.uniform(-VAL_LIMIT, VAL_LIMIT)
        hsv_on = True
        log['hsv'] = (hue, sat, val)
    return M, post, rect, alpha, beta, sigma, seed, hue, sat, val, hsv_on, log


def pack_params(M, alpha=1.0, beta=0.0, sigma=0.0, seed=0, hue=0.0, sat=0.0, val=0.0, hsv_on=False, post=None, rect=None):
    """One row of the kernel's parameter table from a FORWARD homography (source pixel -> output pixel); ``post``: the
    part of it applied after crop + pad, ``rect``: the crop window [x_lo, y_lo, x_hi, y_hi) in the padded frame."""
    row = np.zeros(NPARAM, dtype=np.float32)
    Hinv = np.linalg.inv(np.asarray(M, dtype=np.float64))
    row[0:9] = (Hinv / Hinv[2, 2]).reshape(9)
    Pinv = np.linalg.inv(np.asarray(post if post is not None else np.eye(3), dtype=np.float64))
    row[20:29] = (Pinv / Pinv[2, 2]).reshape(9)
    row[29:33] = rect if rect is not None else (-1e9, -1e9, 1e9, 1e9)
    row[9], row[10], row[11] = alpha, beta, sigma
    row[12] = np.array([seed], dtype=np.uint32).view(np.float32)[0]
    row[13], row[14], row[15] = hue, sat, val
    row[16] = 1.0 if hsv_on else 0.0
    return row


def sample_params(B, S, rng=None, return_log=False):
    rng = rng if rng is not None else np.random.default_rng()
    rows, logs = [], []
    for _ in range(B):
        M, post, rect, alpha, beta, sigma, seed, hue, sat, val, hsv_on, log = sample_frame(S, rng)
        rows.append(pack_params(M, alpha, beta, sigma, seed, hue, sat, val, hsv_on, post, rect))
        logs.append(log)
    out = np.stack(rows)
    return (out, logs) if return_log else out


def augment(img, mask, params):
    """img [B,3,H,W], mask [B,C,H,W]: float32 CUDA tensors; params: numpy / tensor [B, NPARAM]."""
    if not (img.is_cuda and mask.is_cuda and img.dtype == torch.float32 and mask.dtype == torch.float32):
        raise ValueError('augment() needs float32 CUDA tensors')
    B, _, H, W = img.shape
    if mask.shape[0] != B or tuple(mask.shape[2:]) != (H, W):
        raise ValueError(f'mask {tuple(mask.shape)} does not match image {tuple(img.shape)}')
    p = torch.as_tensor(np.asarray(params, dtype=np.float32) if not torch.is_tensor(params) else params, dtype=torch.float32).to(img.device)
    if tuple(p.shape) != (B, NPARAM):
        raise ValueError(f'params must be [{B}, {NPARAM}], got {tuple(p.shape)}')
    img, mask, p = img.contiguous(), mask.contiguous(), p.contiguous()
    img_out, mask_out = torch.empty_like(img), torch.empty_like(mask)
    L.check(L.lib().octseg_augment(L.ptr(img), L.ptr(mask), L.ptr(img_out), L.ptr(mask_out), L.ptr(p), B, mask.shape[1], H, W,
                                   L.stream_ptr()))
    return img_out, mask_out
