// augment.hip -- on-GPU training augmentation (reference src/models/smp/dataset.py:160-207).
//
// The reference augments every frame on the CPU with eight albumentations transforms, each resampling or re-quantising
// the uint8 image in turn.  Here the host draws the same random decisions and parameters per frame
// (oct_segmentation_amd/augment.py mirrors the probabilities and ranges), composes every geometric transform
// (HorizontalFlip, ShiftScaleRotate, RandomCrop + centred PadIfNeeded, Perspective) into ONE inverse homography, and a
// single kernel produces the augmented frame: one bilinear gather of the image (constant-0 border), one nearest gather
// of every mask channel, then GaussNoise, RandomBrightnessContrast and HueSaturationValue on the pixel, each quantised to
// the uint8 grid the way the reference's stage does it (round / truncate / OpenCV's 8-bit HSV arithmetic).  One interpolation instead of up to three: statistical, not
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

// OpenCV's 8-bit RGB <-> HSV as albumentations' HueSaturationValue runs it on uint8 frames (cv2.cvtColor COLOR_RGB2HSV ->
// integer LUT shifts -> COLOR_HSV2RGB; albumentations 1.4.3 _shift_hsv_uint8).  Forward = imgproc color_hsv RGB2HSV_b: integer
// arithmetic, H in [0, 180) and S, V in [0, 255] quantised to 8 bits, divisions through 12-bit fixed-point reciprocal tables
// (sdiv_table[v] = round(255 * 4096 / v), hdiv_table180[d] = round(180 * 4096 / (6 d))).  c0 / c1 / c2 are the "R" / "G" / "B" of
// that call: the reference hands its BGR frames to albumentations, which takes channel 0 for red -- mirrored here on purpose.
__device__ __forceinline__ void rgb2hsv_u8(int r, int g, int b, int& h, int& s, int& v) {
  v = max(r, max(g, b));
  const int vmin = min(r, min(g, b)), diff = v - vmin;
  const int sdiv = v > 0 ? __double2int_rn((255.0 * 4096.0) / (double)v) : 0;
  const int hdiv = diff > 0 ? __double2int_rn((180.0 * 4096.0) / (6.0 * (double)diff)) : 0;
  s = (diff * sdiv + 2048) >> 12;
  int hh;
  if (v == r) hh = g - b;
  else if (v == g) hh = b - r + 2 * diff;
  else hh = r - g + 4 * diff;
  hh = (hh * hdiv + 2048) >> 12;
  if (hh < 0) hh += 180;
  h = hh;
}
// Backward = HSV2RGB_b: through float (H * 6/180, S / 255, V / 255), the sector table of HSV2RGB_native, saturate_cast<uchar>(x * 255)
__device__ __forceinline__ void hsv2rgb_u8(int h, int s, int v, int& r, int& g, int& b) {
  const float fs = (float)s * (1.f / 255.f), fv = (float)v * (1.f / 255.f);
  float fr, fg, fb;
  if (s == 0) { fr = fg = fb = fv; }
  else {
    float hh = (float)h * (6.f / 180.f);
    if (hh < 0.f) { do hh += 6.f; while (hh < 0.f); }
    else if (hh >= 6.f) { do hh -= 6.f; while (hh >= 6.f); }
    int sector = (int)floorf(hh);
    hh -= (float)sector;
    if ((unsigned)sector >= 6u) { sector = 0; hh = 0.f; }
    const float tab[4] = {fv, fv * (1.f - fs), fv * (1.f - fs * hh), fv * (1.f - fs * (1.f - hh))};
    // sector_data[sector] = indices of (b, g, r) in tab
    const int sd[6][3] = {{1, 3, 0}, {1, 0, 2}, {3, 0, 1}, {0, 2, 1}, {0, 1, 3}, {2, 1, 0}};
    fb = tab[sd[sector][0]]; fg = tab[sd[sector][1]]; fr = tab[sd[sector][2]];
  }
  r = min(max(__float2int_rn(fr * 255.f), 0), 255);
  g = min(max(__float2int_rn(fg * 255.f), 0), 255);
  b = min(max(__float2int_rn(fb * 255.f), 0), 255);
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
    // ---- GaussNoise (per channel), RandomBrightnessContrast, HueSaturationValue on the 0..255 pixel.  The reference's frames are
    // uint8 between the transforms: a warp rounds to nearest (cv2 fixed-point interpolation), GaussNoise and
    // RandomBrightnessContrast clip and TRUNCATE (albumentations `np.clip(...).astype(uint8)`), HueSaturationValue runs OpenCV's 8-bit
    // HSV round trip -- each stage is quantised the same way here
#pragma unroll
    for (int c = 0; c < 3; ++c) px[c] = rintf(fminf(fmaxf(px[c], 0.f), 255.f));
    const float sigma = p[11];
    if (sigma > 0.f) {
      const unsigned seed = __float_as_uint(p[12]);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float u1 = u01(hash3(seed, (unsigned)i, 2u * c)), u2 = u01(hash3(seed, (unsigned)i, 2u * c + 1u));
        px[c] = floorf(fminf(fmaxf(px[c] + sigma * sqrtf(-2.f * logf(u1)) * cosf(6.28318530718f * u2), 0.f), 255.f));
      }
    }
    if (p[9] != 1.f || p[10] != 0.f) {   // the LUT of _brightness_contrast_adjust_uint: float32 arange * alpha + beta * 255, clipped, truncated
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float l = px[c];
        if (p[9] != 1.f) l *= p[9];
        if (p[10] != 0.f) l += p[10] * 255.f;
        px[c] = floorf(fminf(fmaxf(l, 0.f), 255.f));
      }
    }
    if (((int)p[16]) & 1) {
      int h, sa, v, r8, g8, b8;
      rgb2hsv_u8((int)px[0], (int)px[1], (int)px[2], h, sa, v);     // channel 0 is albumentations' "R" (see above)
      // LUTs of _shift_hsv_uint8: (arange(256, int16) + shift) mod 180 / clipped, cast to uint8: the shifts are Python floats, the
      // float sum is truncated by the cast
      const float hs = p[13], ss = p[14], vs = p[15];
      if (hs != 0.f) { float t = fmodf((float)h + hs, 180.f); if (t < 0.f) t += 180.f; h = (int)t; }   // np.mod: result has the divisor's sign
      if (ss != 0.f) sa = (int)fminf(fmaxf((float)sa + ss, 0.f), 255.f);
      if (vs != 0.f) v = (int)fminf(fmaxf((float)v + vs, 0.f), 255.f);
      hsv2rgb_u8(h, sa, v, r8, g8, b8);
      px[0] = (float)r8; px[1] = (float)g8; px[2] = (float)b8;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) img_out[((size_t)n * 3 + c) * H * W + (size_t)y * W + x] = px[c];
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
