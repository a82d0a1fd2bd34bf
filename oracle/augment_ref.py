"""Oracle restatement of the reference's training augmentation chain (TEST INFRASTRUCTURE ONLY).

Reference: ``OCTDataset.get_img_augmentation`` (``src/models/smp/dataset.py:160-207``) = albumentations 1.4.3
(``environment.yaml:11``) Compose of HorizontalFlip, ShiftScaleRotate, RandomCrop + PadIfNeeded, GaussNoise, Perspective,
RandomBrightnessContrast, HueSaturationValue, applied to uint8 BGR frames (``cv2.imread``) -- every stage hands a uint8 image
to the next.  albumentations and cv2 are absent here and not under /root/reference, so their published algorithms are
restated in numpy, stage by stage, each quantised to uint8 as the original does:

  * geometric stages: float bilinear resampling + round to nearest (OpenCV's warpAffine / warpPerspective interpolate in
    1/32-pixel fixed point: statistically, not bit, equal), constant-0 border, masks nearest;
  * GaussNoise / RandomBrightnessContrast: ``np.clip(...).astype(uint8)`` = clip and TRUNCATE (albumentations ``clip`` /
    ``_brightness_contrast_adjust_uint`` LUT with ``brightness_by_max``);
  * HueSaturationValue: ``_shift_hsv_uint8`` = cv2 COLOR_RGB2HSV (8-bit: integer arithmetic with 12-bit reciprocal tables,
    H in [0, 180)) -> integer LUT shifts -> COLOR_HSV2RGB (through float, sector table of HSV2RGB_native).  The frames are BGR
    but albumentations assumes RGB: channel 0 plays "R" -- kept.

Parity unpinned by the reference (no augmentation fixtures exist); the GPU kernel (csrc/augment.hip) is tested against this
chain: exactly for the photometric stages (<= 1 grey level), statistically for the resampled ones.
"""
import numpy as np

_SECTOR = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])   # (b, g, r) indices into tab


def rgb2hsv_u8(img):
    """cv2.cvtColor(img_u8, COLOR_RGB2HSV): RGB2HSV_b with hrange 180, hsv_shift 12."""
    x = img.astype(np.int64)
    r, g, b = x[..., 0], x[..., 1], x[..., 2]
    v = np.maximum(r, np.maximum(g, b))
    vmin = np.minimum(r, np.minimum(g, b))
    diff = v - vmin
    sdiv = np.zeros(256, dtype=np.int64); hdiv = np.zeros(256, dtype=np.int64)
    i = np.arange(1, 256, dtype=np.float64)
    sdiv[1:] = np.rint((255 << 12) / (1.0 * i)).astype(np.int64)
    hdiv[1:] = np.rint((180 << 12) / (6.0 * i)).astype(np.int64)
    s = (diff * sdiv[v] + (1 << 11)) >> 12
    h = np.where(v == r, g - b, np.where(v == g, b - r + 2 * diff, r - g + 4 * diff))
    h = (h * hdiv[diff] + (1 << 11)) >> 12
    h = h + np.where(h < 0, 180, 0)
    return np.stack([h, s, v], axis=-1).astype(np.uint8)


def hsv2rgb_u8(hsv):
    """cv2.cvtColor(hsv_u8, COLOR_HSV2RGB): HSV2RGB_b -> float HSV2RGB_native -> saturate_cast<uchar>(x * 255)."""
    h = hsv[..., 0].astype(np.float32) * np.float32(6.0 / 180.0)
    s = hsv[..., 1].astype(np.float32) * np.float32(1.0 / 255.0)
    v = hsv[..., 2].astype(np.float32) * np.float32(1.0 / 255.0)
    h = np.where(h >= 6, h - 6, h)
    sector = np.floor(h).astype(np.int64)
    f = (h - sector.astype(np.float32)).astype(np.float32)
    bad = (sector < 0) | (sector >= 6)
    sector = np.where(bad, 0, sector); f = np.where(bad, np.float32(0), f)
    one = np.float32(1.0)
    tab = np.stack([v, v * (one - s), v * (one - s * f), v * (one - s * (one - f))], axis=-1)
    idx = _SECTOR[sector]                                  # [..., 3] = (b, g, r)
    bgr = np.take_along_axis(tab, idx, axis=-1)
    gray = hsv[..., 1] == 0
    bgr = np.where(gray[..., None], v[..., None], bgr)
    out = np.clip(np.rint(bgr * np.float32(255.0)), 0, 255).astype(np.uint8)
    return out[..., ::-1]                                  # -> (r, g, b)


def shift_hsv_uint8(img, hue_shift, sat_shift, val_shift):
    """albumentations 1.4.3 functional._shift_hsv_uint8 (img is what albumentations believes to be RGB)."""
    hsv = rgb2hsv_u8(img)
    hue, sat, val = hsv[..., 0], hsv[..., 1], hsv[..., 2]
    if hue_shift != 0:
        lut = np.mod(np.arange(0, 256, dtype=np.int16) + hue_shift, 180).astype(np.uint8)
        hue = lut[hue]
    if sat_shift != 0:
        lut = np.clip(np.arange(0, 256, dtype=np.int16) + sat_shift, 0, 255).astype(np.uint8)
        sat = lut[sat]
    if val_shift != 0:
        lut = np.clip(np.arange(0, 256, dtype=np.int16) + val_shift, 0, 255).astype(np.uint8)
        val = lut[val]
    return hsv2rgb_u8(np.stack([hue, sat, val], axis=-1))


def brightness_contrast_uint8(img, alpha, beta):
    """_brightness_contrast_adjust_uint with beta_by_max=True (RandomBrightnessContrast default)."""
    lut = np.arange(0, 256).astype('float32')
    if alpha != 1:
        lut *= np.float32(alpha)
    if beta != 0:
        lut += np.float32(beta * 255)
    lut = np.clip(lut, 0, 255).astype(np.uint8)
    return lut[img]


def gauss_noise_uint8(img, gauss):
    """gauss_noise: img.astype(float32) + gauss, then the @clipped wrapper: np.clip(..., 0, 255).astype(uint8)."""
    return np.clip(img.astype('float32') + gauss.astype('float32'), 0, 255).astype(np.uint8)


def warp(img, M, nearest=False):
    """dst(x, y) = src(M^-1 (x, y)): bilinear (or nearest) with constant-0 border, uint8 in / out; M maps source -> output pixels."""
    H, W = img.shape[:2]
    Hi = np.linalg.inv(np.asarray(M, dtype=np.float64))
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing='ij')
    w = Hi[2, 0] * xs + Hi[2, 1] * ys + Hi[2, 2]
    sx = (Hi[0, 0] * xs + Hi[0, 1] * ys + Hi[0, 2]) / w
    sy = (Hi[1, 0] * xs + Hi[1, 1] * ys + Hi[1, 2]) / w
    src = img.astype(np.float64)
    if src.ndim == 2:
        src = src[..., None]

    def at(yy, xx):
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        return np.where(ok[..., None], src[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)], 0.0)
    if nearest:
        out = at(np.floor(sy + 0.5).astype(np.int64), np.floor(sx + 0.5).astype(np.int64))
    else:
        x0, y0 = np.floor(sx).astype(np.int64), np.floor(sy).astype(np.int64)
        ax, ay = (sx - x0)[..., None], (sy - y0)[..., None]
        out = (at(y0, x0) * (1 - ax) + at(y0, x0 + 1) * ax) * (1 - ay) + (at(y0 + 1, x0) * (1 - ax) + at(y0 + 1, x0 + 1) * ax) * ay
        out = np.rint(out)
    out = np.clip(out, 0, 255).astype(img.dtype)
    return out if img.ndim == 3 else out[..., 0]


def apply_chain(img, mask, log, S, gauss=None):
    """The sequential chain on one frame.  img: uint8 [S, S, 3] (BGR as read by cv2), mask: uint8 [S, S, C] of 0/1,
    log: the decisions of oct_segmentation_amd.augment.sample_frame (its last return value), gauss: noise field [S, S, 3]."""
    c = S / 2 - 0.5
    T = lambda tx, ty: np.array([[1, 0, tx], [0, 1, ty], [0, 0, 1]], dtype=np.float64)   # noqa: E731
    if log.get('hflip'):
        img, mask = img[:, ::-1].copy(), mask[:, ::-1].copy()
    if 'ssr' in log:
        angle, scale, dx, dy = log['ssr']
        a = np.deg2rad(angle)
        R = np.array([[scale * np.cos(a), scale * np.sin(a), 0], [-scale * np.sin(a), scale * np.cos(a), 0], [0, 0, 1]])
        M = T(dx, dy) @ T(c, c) @ R @ T(-c, -c)
        img, mask = warp(img, M), warp(mask, M, nearest=True)
    if 'crop' in log:
        y0, x0, ch, cw, pad_t, pad_l = log['crop']
        ci, cm = img[y0:y0 + ch, x0:x0 + cw], mask[y0:y0 + ch, x0:x0 + cw]
        img = np.zeros_like(img); mask = np.zeros_like(mask)
        img[pad_t:pad_t + ch, pad_l:pad_l + cw] = ci
        mask[pad_t:pad_t + ch, pad_l:pad_l + cw] = cm
    if 'noise' in log and gauss is not None:
        img = gauss_noise_uint8(img, gauss)
    if 'perspective_matrix' in log:
        P = log['perspective_matrix']
        img, mask = warp(img, P), warp(mask, P, nearest=True)
    if 'bc' in log:
        img = brightness_contrast_uint8(img, *log['bc'])
    if 'hsv' in log:
        img = shift_hsv_uint8(img, *log['hsv'])
    return img, mask
