// augment.hip -- on-GPU training augmentation (reference src/models/smp/dataset.py:160-207).
//
// The reference augments every frame on the CPU with eight albumentations transforms, each resampling or re-quantising
// the uint8 image in turn.  Here the host draws the same random decisions and parameters per frame
// (oct_segmentation_amd/augment.py mirrors the probabilities and ranges), composes every geometric transform
// (HorizontalFlip, ShiftScaleRotate, RandomCrop + centred PadIfNeeded, Perspective) into ONE inverse homography, and a
// single kernel produces the augmented frame: one bilinear gather of the image (constant-0 border), one nearest gather
// of every mask channel, then GaussNoise, RandomBrightnessContrast and HueSaturationValue on the pixel, clipped and
// rounded to the uint8 grid the reference's images live on.  One interpolation instead of up to three: statistical, not
// bit, parity with the reference -- which is what section 8 asks of this row.
#include "common.h"
#include "kernels.h"

namespace octseg {

namespace {

__device__ __forceinline__ unsigned hash3(unsigned a, unsigned b, unsigned c) {   // small counter-based generator (lowbias32 rounds)
  unsigned x = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u ^ (c + 0x165667B1u) * 0xC2B2AE3Du;
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ float u01(unsigned h) { return ((float)(h >> 8) + 0.5f) * (1.0f / 16777216.0f); }

// OpenCV's 8-bit BGR <-> HSV (H in [0, 180), S and V in [0, 255]), in float
__device__ __forceinline__ void bgr2hsv(float b, float g, float r, float& h, float& s, float& v) {
  v = fmaxf(r, fmaxf(g, b));
  const float mn = fminf(r, fminf(g, b)), d = v - mn;
  s = v > 0.f ? 255.f * d / v : 0.f;
  if (d <= 0.f) { h = 0.f; return; }
  float hh;
  if (v == r) hh = (g - b) / d;
  else if (v == g) hh = 2.f + (b - r) / d;
  else hh = 4.f + (r - g) / d;
  hh *= 30.f;                       // 60 degrees / 2
  if (hh < 0.f) hh += 180.f;
  h = hh;
}
__device__ __forceinline__ void hsv2bgr(float h, float s, float v, float& b, float& g, float& r) {
  const float hh = h / 30.f, ss = s / 255.f;
  const int sector = ((int)floorf(hh)) % 6;
  const float f = hh - floorf(hh);
  const float p = v * (1.f - ss), q = v * (1.f - ss * f), t = v * (1.f - ss * (1.f - f));
  switch (sector) {
    case 0: r = v; g = t; b = p; break;
    case 1: r = q; g = v; b = p; break;
    case 2: r = p; g = v; b = t; break;
    case 3: r = p; g = q; b = v; break;
    case 4: r = t; g = p; b = v; break;
    default: r = v; g = p; b = q; break;
  }
}

}  // namespace

// params[n][AUG_NPARAM]: 0-8 inverse homography (output pixel -> source pixel, row major), 9 contrast alpha, 10 brightness
// beta (added as beta * 255), 11 noise sigma (0 = none), 12 seed (as float bits), 13 hue shift (degrees / 2, OpenCV units),
// 14 saturation shift, 15 value shift, 16 flags (bit 0: apply HSV), 17-19 reserved, 20-28 inverse homography output pixel ->
// frame after RandomCrop + PadIfNeeded (identity without Perspective), 29-32 the crop window in that frame [x_lo, y_lo, x_hi, y_hi):
// samples that fall outside it are the padding (0) -- a crop is a shift PLUS this blanking
__global__ __launch_bounds__(256) void augment_kernel(const float* img, const float* mask, float* img_out, float* mask_out,
                                                      const float* params, int B, int C, int H, int W) {
  const size_t total = (size_t)B * H * W;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % W);
    const int y = (int)((i / W) % H);
    const int n = (int)(i / ((size_t)W * H));
    const float* p = params + (size_t)n * AUG_NPARAM;
    const float sw = p[6] * x + p[7] * y + p[8];
    const float isw = sw != 0.f ? 1.f / sw : 0.f;
    const float sx = (p[0] * x + p[1] * y + p[2]) * isw, sy = (p[3] * x + p[4] * y + p[5]) * isw;
    // the crop window, tested where the reference would have padded: in the frame between crop+pad and perspective
    const float cw_ = p[26] * x + p[27] * y + p[28];
    const float icw = cw_ != 0.f ? 1.f / cw_ : 0.f;
    const float cx = (p[20] * x + p[21] * y + p[22]) * icw, cy = (p[23] * x + p[24] * y + p[25]) * icw;
    const bool in_crop = cx >= p[29] - 0.5f && cy >= p[30] - 0.5f && cx < p[31] - 0.5f && cy < p[32] - 0.5f;
    // ---- image: bilinear, constant 0 outside
    const float fx = floorf(sx), fy = floorf(sy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float ax = sx - fx, ay = sy - fy;
    float px[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* pl = img + ((size_t)n * 3 + c) * H * W;
      auto at = [&](int yy, int xx) { return (yy >= 0 && yy < H && xx >= 0 && xx < W) ? pl[(size_t)yy * W + xx] : 0.f; };
      const float top = at(y0, x0) * (1.f - ax) + at(y0, x0 + 1) * ax;
      const float bot = at(y0 + 1, x0) * (1.f - ax) + at(y0 + 1, x0 + 1) * ax;
      px[c] = in_crop ? top * (1.f - ay) + bot * ay : 0.f;
    }
    // ---- GaussNoise (per channel), RandomBrightnessContrast, HueSaturationValue: on the 0..255 BGR pixel
    const float sigma = p[11];
    if (sigma > 0.f) {
      const unsigned seed = __float_as_uint(p[12]);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float u1 = u01(hash3(seed, (unsigned)i, 2u * c)), u2 = u01(hash3(seed, (unsigned)i, 2u * c + 1u));
        px[c] += sigma * sqrtf(-2.f * logf(u1)) * cosf(6.28318530718f * u2);
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) px[c] = fminf(fmaxf(px[c] * p[9] + p[10] * 255.f, 0.f), 255.f);
    if (((int)p[16]) & 1) {
      float h, s, v;
      bgr2hsv(px[0], px[1], px[2], h, s, v);
      h = fmodf(h + p[13] + 360.f, 180.f);
      s = fminf(fmaxf(s + p[14], 0.f), 255.f);
      v = fminf(fmaxf(v + p[15], 0.f), 255.f);
      hsv2bgr(h, s, v, px[0], px[1], px[2]);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) img_out[((size_t)n * 3 + c) * H * W + (size_t)y * W + x] = rintf(fminf(fmaxf(px[c], 0.f), 255.f));
    // ---- mask: nearest, 0 outside
    const int mx = (int)floorf(sx + 0.5f), my = (int)floorf(sy + 0.5f);
    const bool inside = in_crop && mx >= 0 && mx < W && my >= 0 && my < H;
    for (int c = 0; c < C; ++c)
      mask_out[((size_t)n * C + c) * H * W + (size_t)y * W + x] = inside ? mask[((size_t)n * C + c) * H * W + (size_t)my * W + mx] : 0.f;
  }
}

hipError_t launch_augment(const float* img, const float* mask, float* img_out, float* mask_out, const float* params, int B, int C, int H,
                          int W, hipStream_t st) {
  const size_t total = (size_t)B * H * W;
  size_t g = (total + 255) / 256;
  if (g > 16384) g = 16384;
  hipLaunchKernelGGL(augment_kernel, dim3((unsigned)g), dim3(256), 0, st, img, mask, img_out, mask_out, params, B, C, H, W);
  return hipGetLastError();
}

}  // namespace octseg
