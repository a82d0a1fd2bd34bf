"""Oracle restatement of the two OpenCV resizes on the predict path (TEST INFRASTRUCTURE ONLY).

The reference calls ``cv2.resize(predict_mask, tuple(output_size), interpolation=cv2.INTER_NEAREST)``
(``src/predict.py:92-96``) and ``cv2.resize(image, (input_size, input_size))`` on uint8 BGR frames
(``src/data/utils.py:159-166``).  ``opencv-python==4.8.1.78`` (``environment.yaml:23``) is absent from this image
and its source is not under /root/reference, so the algorithms of ``modules/imgproc/src/resize.cpp`` are restated
here as scalar loops in the order of the C++ (``resizeNN`` / ``resizeNNInvoker``; ``resize_`` coefficient set-up,
``HResizeLinear<uchar,int,short,2048>``, ``VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>>``).
Parity unpinned by the reference (it holds no resize fixture); the product's vectorised tables
(``oct_segmentation_amd/predict.py``) are tested against these loops.
"""
import math
import struct

import numpy as np


def _f32(x):
    """Round a Python float to the nearest IEEE float32 (the C++ code keeps fx / the coefficients in float)."""
    return struct.unpack('f', struct.pack('f', x))[0]


def resize_nn(src, dsize):
    """resizeNN: dsize = (width, height); src [H, W] or [H, W, C] of any dtype."""
    src = np.asarray(src)
    dw, dh = int(dsize[0]), int(dsize[1])
    sh, sw = src.shape[:2]
    fx, fy = dw / float(sw), dh / float(sh)          # inv_scale_x / inv_scale_y of cv::resize (double)
    ifx, ify = 1.0 / fx, 1.0 / fy
    x_ofs = [min(int(math.floor(x * ifx)), sw - 1) for x in range(dw)]
    out = np.empty((dh, dw) + src.shape[2:], dtype=src.dtype)
    for y in range(dh):
        sy = min(int(math.floor(y * ify)), sh - 1)
        for x in range(dw):
            out[y, x] = src[sy, x_ofs[x]]
    return out


def _linear_tabs(ssize, dsize, horizontal):
    """(ofs, alpha[2]) per destination coordinate, as resize_() builds them for INTER_LINEAR with a fixed-point kernel.
    Only the horizontal table zeroes the fraction when the left tap leaves the row (``fx = 0, sx = 0`` / ``sx = width - 1``);
    the vertical one keeps (1 - fy, fy) and resizeGeneric_Invoker clips the two row indices instead."""
    scale = 1.0 / (dsize / float(ssize))
    ofs, alpha = [], []
    for d in range(dsize):
        f = _f32((d + 0.5) * scale - 0.5)
        s = int(math.floor(f))
        f = _f32(f - s)
        if horizontal and s < 0:
            f, s = 0.0, 0
        if horizontal and s >= ssize - 1:
            f, s = 0.0, ssize - 1
        c0, c1 = _f32(1.0 - f), f
        # saturate_cast<short>(float) = cvRound = round half to even
        a0 = int(np.rint(np.float32(c0) * np.float32(2048.0)))
        a1 = int(np.rint(np.float32(c1) * np.float32(2048.0)))
        ofs.append(s)
        alpha.append((a0, a1))
    return ofs, alpha


def resize_linear_u8(src, dsize):
    """cv2.resize(src_u8, dsize) with the default INTER_LINEAR: dsize = (width, height)."""
    src = np.asarray(src)
    assert src.dtype == np.uint8
    if src.ndim == 2:
        return resize_linear_u8(src[:, :, None], dsize)[:, :, 0]
    dw, dh = int(dsize[0]), int(dsize[1])
    sh, sw, cn = src.shape
    # cv::resize: scale_x = 1. / inv_scale_x; iscale_x = saturate_cast<int>(scale_x); is_area_fast = both |scale - iscale| < DBL_EPSILON;
    # `if (interpolation == INTER_LINEAR && is_area_fast && iscale_x == 2 && iscale_y == 2) interpolation = INTER_AREA;` and the 8-bit
    # resizeAreaFast_ with ResizeAreaFastVec (fast_mode: scale 2x2, cn in {1, 3, 4}): D = (S[i] + S[i + cn] + nextS[i] + nextS[i + cn] + 2) >> 2
    scale_x, scale_y = 1.0 / (dw / float(sw)), 1.0 / (dh / float(sh))
    eps = 2.220446049250313e-16
    if int(round(scale_x)) == 2 and int(round(scale_y)) == 2 and abs(scale_x - 2) < eps and abs(scale_y - 2) < eps and cn in (1, 3, 4):
        out = np.empty((dh, dw, cn), dtype=np.uint8)
        for y in range(dh):
            for x in range(dw):
                for c in range(cn):
                    out[y, x, c] = (int(src[2 * y, 2 * x, c]) + int(src[2 * y, 2 * x + 1, c]) + int(src[2 * y + 1, 2 * x, c]) +
                                    int(src[2 * y + 1, 2 * x + 1, c]) + 2) >> 2
        return out
    xofs, xa = _linear_tabs(sw, dw, True)
    yofs, ya = _linear_tabs(sh, dh, False)
    out = np.empty((dh, dw, cn), dtype=np.uint8)
    for y in range(dh):
        sy0 = min(max(yofs[y], 0), sh - 1)          # clip(sy0 - ksize2 + 1 + k, 0, ssize.height), k = 0, 1
        sy1 = min(max(yofs[y] + 1, 0), sh - 1)
        b0, b1 = ya[y]
        for x in range(dw):
            sx0 = xofs[x]
            sx1 = min(sx0 + 1, sw - 1)
            a0, a1 = xa[x]
            for c in range(cn):
                r0 = int(src[sy0, sx0, c]) * a0 + int(src[sy0, sx1, c]) * a1      # HResizeLinear, int rows
                r1 = int(src[sy1, sx0, c]) * a0 + int(src[sy1, sx1, c]) * a1
                v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2
                out[y, x, c] = min(max(v, 0), 255)
    return out
