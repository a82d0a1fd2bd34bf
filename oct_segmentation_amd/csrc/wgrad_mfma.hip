// wgrad_mfma.hip -- weight gradients of the NHWC convolutions on gfx950 matrix cores.
//
//   dW[tap][co][ci] = sum_pixels dy[pixel][co] * xin[pixel*istride + tap][ci]
//
// is a GEMM whose contraction index (pixels) is the *slow* axis of both NHWC operands.
// bf16: both tiles are staged pixel-major in LDS exactly as they lie in HBM and the MFMA
// operands are fetched with ds_read_b64_tr_b16 (hardware transpose read), so no transposed
// copy of dy or of the activations ever exists.  f32: v_mfma_f32_32x32x2_f32 takes one
// element per lane, a plain ds_read_b32 column read is already conflict free.
// The input window is the same virtual tensor as in the forward (concat / nearest-x2 /
// lazy BN+ReLU applied while staging), and all taps of one (co,ci) tile are accumulated
// from a single staged window: accumulators = NTAPS x 32x32 per wave.
// Workgroup = 4 waves = a 64(co) x 64(ci) x NTAPS tile; grid.z splits the pixel tiles and
// partial results are combined with fp32 atomics shaped as 128-byte row segments.
#include "common.h"
#include "conv_common.h"
#include "kernels.h"

namespace octseg {

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;

template <typename T, int NTAPS>
__global__ __launch_bounds__(NTHR) void wgrad_mfma_kernel(const WgradArgs a, const int th) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int VEC = Tr<T>::VEC;
  constexpr int RB = 64 * (int)sizeof(T);  // 64 channels per LDS row
  constexpr int PITCH = RB + 16;
  constexpr int VPR = RB / 16;
  constexpr int PSTEP = NTHR / VPR;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wq_m = wave >> 1, wq_n = wave & 1;
  const int ci0 = blockIdx.x * 64, co0 = blockIdx.y * 64;

  const int tiles_x = (a.OW + TW - 1) / TW, tiles_y = (a.OH + th - 1) / th;
  const int ntiles = a.N * tiles_x * tiles_y;

  const bool single = a.ntaps == 1;
  const int lstride = single ? 1 : a.istride;
  const int smul = single ? a.istride : 1;
  const int RH = single ? th : (th - 1) * a.istride + a.span_y;
  const int RW = single ? TW : (TW - 1) * a.istride + a.span_x;
  const int npix = RH * RW;
  const float inv_rw = 1.0f / (float)RW;

  char* ldsY = smem;                          // [th*16][64 ch] dy tile
  char* ldsX = smem + th * TW * PITCH;        // [RH*RW][64 ch] input window

  f32x16_t acc[NTAPS];
#pragma unroll
  for (int t = 0; t < NTAPS; ++t)
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[t][k] = 0.f;

  int toff[NTAPS];
#pragma unroll
  for (int t = 0; t < NTAPS; ++t)
    toff[t] = single ? 0 : ((a.tap_dy[t] - a.min_dy) * RW + (a.tap_dx[t] - a.min_dx)) * PITCH;

  // per-lane operand addressing
  int ya0, xa0, xrow_step;  // dy-tile / window byte offsets for pixel tx=lane-dependent part, row kk = 0
  if constexpr (sizeof(T) == 2) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3, hh = g >> 1, cb = g & 1;
    const int tx = 8 * hh + q;
    ya0 = tx * PITCH + (wq_m * 32 + 16 * cb + 4 * p) * 2;
    xa0 = (tx * lstride) * PITCH + (wq_n * 32 + 16 * cb + 4 * p) * 2;
    xrow_step = 4 * lstride * PITCH;  // second transposed read: 4 pixels further
  } else {
    const int r = lane & 31, h = lane >> 5;
    ya0 = h * PITCH + (wq_m * 32 + r) * 4;
    xa0 = (h * lstride) * PITCH + (wq_n * 32 + r) * 4;
    xrow_step = 2 * lstride * PITCH;  // next MFMA: 2 pixels further
  }

  for (int tile = blockIdx.z; tile < ntiles; tile += a.ksplit) {
    int rem = tile;
    const int n = rem / (tiles_x * tiles_y);
    rem -= n * tiles_x * tiles_y;
    const int tyi = rem / tiles_x, txi = rem - tyi * tiles_x;
    const int y0 = tyi * th, x0 = txi * TW;
    __syncthreads();  // previous tile fully consumed
    // ---- stage dy tile (zero outside the grid / channel range) ----
    {
      const int cv = tid % VPR;
      const int c = co0 + cv * VEC;
      for (int p = tid / VPR; p < th * TW; p += PSTEP) {
        const int ty = p >> 4, tx = p & 15;
        const int gy = y0 + ty, gx = x0 + tx;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (gy < a.OH && gx < a.OW && c < a.dyC) {
          const size_t e = (((size_t)n * a.DH + gy * a.dstride + a.doy) * a.DW + gx * a.dstride + a.dox) * a.dyC + c;
          v = *(const uint4*)((const char*)a.dy + e * sizeof(T));
        }
        *(uint4*)(ldsY + p * PITCH + cv * 16) = v;
      }
    }
    // ---- stage the input window (64 channels starting at ci0) ----
    stage_window<T, RB>(ldsX, a.src, a.nsrc, a.Cin, blockIdx.x, n, y0 * a.istride + a.min_dy,
                        x0 * a.istride + a.min_dx, smul, RW, npix, inv_rw, a.IH, a.IW, tid);
    __syncthreads();
    // ---- MFMA over the pixels of the tile ----
    for (int kk = 0; kk < th; ++kk) {
      const char* yrow = ldsY + kk * TW * PITCH + ya0;
      const char* xrow = ldsX + (kk * lstride) * RW * PITCH + xa0;
      if constexpr (sizeof(T) == 2) {
        s16x4_t y_lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(yrow));
        s16x4_t y_hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(yrow + 4 * PITCH));
        uint4 af;
        af.x = ((unsigned)(unsigned short)y_lo[0]) | ((unsigned)(unsigned short)y_lo[1] << 16);
        af.y = ((unsigned)(unsigned short)y_lo[2]) | ((unsigned)(unsigned short)y_lo[3] << 16);
        af.z = ((unsigned)(unsigned short)y_hi[0]) | ((unsigned)(unsigned short)y_hi[1] << 16);
        af.w = ((unsigned)(unsigned short)y_hi[2]) | ((unsigned)(unsigned short)y_hi[3] << 16);
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) {
          s16x4_t x_lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(xrow + toff[t]));
          s16x4_t x_hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(xrow + toff[t] + xrow_step));
          uint4 bf;
          bf.x = ((unsigned)(unsigned short)x_lo[0]) | ((unsigned)(unsigned short)x_lo[1] << 16);
          bf.y = ((unsigned)(unsigned short)x_lo[2]) | ((unsigned)(unsigned short)x_lo[3] << 16);
          bf.z = ((unsigned)(unsigned short)x_hi[0]) | ((unsigned)(unsigned short)x_hi[1] << 16);
          bf.w = ((unsigned)(unsigned short)x_hi[2]) | ((unsigned)(unsigned short)x_hi[3] << 16);
          Tr<T>::mma(af, bf, acc[t]);
        }
      } else {
#pragma unroll
        for (int kp = 0; kp < 8; ++kp) {
          const float yv = *(const float*)(yrow + kp * 2 * PITCH);
#pragma unroll
          for (int t = 0; t < NTAPS; ++t) {
            const float xv = *(const float*)(xrow + toff[t] + kp * xrow_step);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(yv, xv, acc[t], 0, 0, 0);
          }
        }
      }
    }
  }

  // ---- combine: fp32 atomics, lanes 0-31 cover 128 contiguous bytes of one dW row ----
  const int ci = ci0 + wq_n * 32 + (lane & 31);
#pragma unroll
  for (int t = 0; t < NTAPS; ++t) {
    const int tw = a.tap_w[t];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int co = co0 + wq_m * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
      if (co < a.Cout && ci < a.Cin) atomicAdd(a.dW + ((size_t)tw * a.Cout + co) * a.Cin + ci, acc[t][i]);
    }
  }
}

static size_t wgrad_lds_bytes(const WgradArgs& a, int dtype, int th) {
  const int RB = 64 * (int)dtype_size(dtype), PITCH = RB + 16;
  const bool single = a.ntaps == 1;
  const int RH = single ? th : (th - 1) * a.istride + a.span_y;
  const int RW = single ? TW : (TW - 1) * a.istride + a.span_x;
  return (size_t)(th * TW + RH * RW) * PITCH;
}

template <typename T, int NTAPS>
static hipError_t launch_wgrad_t(const WgradArgs& a0, int dtype, hipStream_t st) {
  WgradArgs a = a0;
  int th = 8;
  while (th > 1 && wgrad_lds_bytes(a, dtype, th) > 150 * 1024) th >>= 1;
  const size_t lds = wgrad_lds_bytes(a, dtype, th);
  const int ntiles = a.N * ((a.OW + TW - 1) / TW) * ((a.OH + th - 1) / th);
  const int gx = (a.Cin + 63) / 64, gy = (a.Cout + 63) / 64;
  int ks = (1024 + gx * gy - 1) / (gx * gy);  // aim at ~1024 workgroups
  if (ks > ntiles) ks = ntiles;
  if (ks < 1) ks = 1;
  a.ksplit = ks;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)wgrad_mfma_kernel<T, NTAPS>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL((wgrad_mfma_kernel<T, NTAPS>), dim3(gx, gy, ks), dim3(NTHR), lds, st, a, th);
  return hipGetLastError();
}

// Tap tables with other sizes are split host-side into groups the instantiations cover.
hipError_t launch_wgrad(int dtype, const WgradArgs& a, hipStream_t st) {
  if (a.ntaps <= 0) return hipSuccess;
  if (a.ntaps == 1 || a.ntaps == 4 || a.ntaps == 9) {
    if (dtype == DT_F32) {
      if (a.ntaps == 1) return launch_wgrad_t<float, 1>(a, dtype, st);
      if (a.ntaps == 4) return launch_wgrad_t<float, 4>(a, dtype, st);
      return launch_wgrad_t<float, 9>(a, dtype, st);
    }
    if (a.ntaps == 1) return launch_wgrad_t<bf16_t, 1>(a, dtype, st);
    if (a.ntaps == 4) return launch_wgrad_t<bf16_t, 4>(a, dtype, st);
    return launch_wgrad_t<bf16_t, 9>(a, dtype, st);
  }
  // generic fallback: one tap per launch, window bounding box kept (any tap count, e.g. 7x7 in tests)
  for (int t = 0; t < a.ntaps; ++t) {
    WgradArgs b = a;
    b.ntaps = 1;
    b.tap_dy[0] = a.tap_dy[t]; b.tap_dx[0] = a.tap_dx[t]; b.tap_w[0] = a.tap_w[t];
    b.min_dy = a.tap_dy[t]; b.min_dx = a.tap_dx[t]; b.span_y = 1; b.span_x = 1;
    hipError_t e = dtype == DT_F32 ? launch_wgrad_t<float, 1>(b, dtype, st) : launch_wgrad_t<bf16_t, 1>(b, dtype, st);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

}  // namespace octseg
