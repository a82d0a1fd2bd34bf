// thin.hip -- the thin full-resolution 3x3 layers (<= 32 channels in, <= 32 out) as HBM-bound streaming kernels on gfx950.
//
// Which layers: the last decoder block and the head of U-Net / U-Net++ (reference: smp DecoderBlock conv1 / conv2 at full resolution and
// SegmentationHead, src/models/smp/model.py:38-44 -> smp.create_model; SURVEY Appendix B: x_0_4 32 -> 16 and 16 -> 16 @704^2, head 16 -> C),
// forward, data gradient and weight gradient.  Their arithmetic intensity is 8..72 FLOP/B: the roof is HBM (8 TB/s), not MFMA.
// Through conv_mfma_kernel / wgrad_mfma_kernel (128-byte K chunks, 32..64-channel N tiles, one 16x16-pixel tile per workgroup, a
// 64 x 64 x taps weight-gradient tile) they ran at 68..126 TFLOP/s = 1.2 TB/s of algorithmic bytes, 15 % of the HBM roof (round 2).
//
// Here: a persistent 256-thread workgroup walks 32-pixel-wide tiles; the (tile + halo) window is staged ONCE into LDS with the lazy
// BatchNorm + ReLU (and the nearest-x2 read) applied, and
//   * conv / dgrad (thin_conv_kernel): the WEIGHTS are the MFMA A operand, built once per workgroup from the fp32 master weights and
//     kept in registers for every tile (row = output channel, k = (tap, input channel) pairs packed into 32-wide k-steps: 5 steps for
//     16 channels, 9 for 32); the B operand is 16 pixels of a window row read with one ds_read_b128 per k-step (conflict-free: 32-byte
//     pixels, or 96-byte pitch for 64-byte pixels).  v_mfma_f32_16x16x32 leaves D[row = channel 4g + j][col = pixel] : every lane
//     owns 4 consecutive channels of one pixel = one 8-byte NHWC store, no LDS transpose.  BatchNorm partial sums stay in registers
//     over all tiles of a workgroup (one slab row per workgroup).  The gradient of a nearest-x2 upsample is summed in registers
//     (two rows into one accumulator, lane pairs by DPP) in f32 and rounded once.
//   * wgrad (thin_wgrad_kernel): contraction over pixels, both operands fetched with ds_read_b64_tr_b16 from the pixel-major images
//     (k order permuted so that a 32-lane half reads 8 consecutive pixels = 256 contiguous bytes: conflict-free), 9 x (CIN / 16)
//     accumulators of 16 x 16 per wave kept across all tiles of the workgroup, one LDS reduction + one round of fp32 atomics at the end.
// No packed weight image is read: both kernels take the fp32 master weights ([R][S][O][I]) and round them exactly as pack_all does.
#include "common.h"
#include "conv_common.h"
#include "kernels.h"

#include <cstdlib>
#include <cstring>

namespace octseg {

namespace {

constexpr int THIN_TW = 32;          // tile width in pixels (two 16-pixel MFMA column groups)
constexpr int THIN_RW = THIN_TW + 2; // window width
constexpr int THIN_NTHR = 256;
constexpr int THIN_MAXWG = 768;      // persistent workgroups (3 per CU: 150 registers per lane)

struct ThinArgs {
  const char* x; const float* scale; const float* shift; int relu, up, sH, sW;   // source: NHWC T [N][sH][sW][CIN] (stored extent), lazy BN
  int N, OH, OW;                     // output grid = virtual input extent (3x3, stride 1, pad 1)
  int tdy[9], tdx[9], tw[9];         // window offset (0..2) and master-weight tap of each of the 9 taps
  const float* w; int wO, wI, wtrans; const float* wscale;   // fp32 master [9][wO][wI]; wtrans: rows run over I, contraction over O
  int Cout;                          // output channels (rows of the A operand that are real)
  const float* bias; int relu_out;
  float* slab; int slab_row0;        // BN partial sums: one row per workgroup, [rows][Cout][2]
  char* y; int yC, yH, yW, accum, pool, head;   // destination (NHWC T; head: NCHW f32 [N][Cout][yH][yW])
  int tiles_x, tiles_y, ntiles;
  // wgrad only
  const char* dy; int dyC; float* dW;
};

template <int CIN> struct ThinCfg {
  static constexpr int PIXB = CIN == 16 ? 32 : 96;     // LDS bytes per window pixel (64-byte pixels padded: conflict-free b128 / tr reads)
  static constexpr int VPP = CIN / 8;                  // 16-byte vectors per pixel
  static constexpr int STEPS = (9 * CIN + 31) / 32;    // 32-wide k-steps over the (tap, channel) pairs
};

// Stage the (TH + 2) x 34 window of tile (n, y0, x0) into LDS: lazy BN + ReLU, nearest-x2 read, zero outside the image.
template <typename T, int CIN, int TH>
__device__ __forceinline__ void thin_stage(const ThinArgs& a, char* lds, int n, int y0, int x0, int tid, const float (&sc)[8],
                                           const float (&sh)[8], bool has_aff) {
  constexpr int PIXB = ThinCfg<CIN>::PIXB, VPP = ThinCfg<CIN>::VPP;
  constexpr int NPX = (TH + 2) * THIN_RW, NVEC = NPX * VPP, MAXV = (NVEC + THIN_NTHR - 1) / THIN_NTHR;
  const int cv = tid % VPP;
  const char* img = a.x + ((size_t)n * a.sH * a.sW * CIN + cv * 8) * sizeof(T);
  uint4 v[MAXV];
  bool ok[MAXV];
  int dst[MAXV];
#pragma unroll
  for (int u = 0; u < MAXV; ++u) {
    const int vi = min(tid + u * THIN_NTHR, NVEC - 1);      // (the clamped tail re-stages the last vector: same value, harmless)
    const int px = vi / VPP;
    const int hy = (px * 1928) >> 16;                      // px / 34 for px < 2312
    const int hx = px - hy * THIN_RW;
    const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
    ok[u] = (unsigned)iy < (unsigned)a.OH && (unsigned)ix < (unsigned)a.OW;
    const int iyc = min(max(iy, 0), a.OH - 1) >> a.up, ixc = min(max(ix, 0), a.OW - 1) >> a.up;
    v[u] = *(const uint4*)(img + ((size_t)iyc * a.sW + ixc) * (CIN * sizeof(T)));
    dst[u] = px * PIXB + cv * 16;
  }
#pragma unroll
  for (int u = 0; u < MAXV; ++u) {
    uint4 w = v[u];
    if (has_aff) w = Tr<T>::affine(w, sc, sh, a.relu);
    if (!ok[u]) w = make_uint4(0, 0, 0, 0);
    *(uint4*)(lds + dst[u]) = w;
  }
}

}  // namespace

template <typename T, int CIN, int NB, int TH>
__global__ __launch_bounds__(THIN_NTHR, NB == 1 ? 3 : 2) void thin_conv_kernel(const ThinArgs a) {
  constexpr int PIXB = ThinCfg<CIN>::PIXB, STEPS = ThinCfg<CIN>::STEPS;
  constexpr int RPW = TH / 4;                              // tile rows per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* win = smem;
  float* red = (float*)(smem + (TH + 2) * THIN_RW * PIXB);   // [4 waves][NB * 16][2]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g = lane >> 4;

  // ---- A operand: the weights, rounded to T, in registers for the whole kernel.  Lane (row = r16, k-slice g) of k-step s holds
  // k = 32 s + 8 g + j, j = 0..7  <->  tap k / CIN, contraction channel k % CIN.
  uint4 wf[NB][STEPS];
  int toff[STEPS];
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    const int k0 = 32 * s + 8 * g;
    const int ti = k0 / CIN, c0 = k0 % CIN;
    const bool tv = ti < 9;
    const int tc = tv ? ti : 8;
    toff[s] = (a.tdy[tc] * THIN_RW + a.tdx[tc]) * PIXB + c0 * (int)sizeof(T);
    const int tw = a.tw[tc];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int row = b * 16 + r16;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = c0 + j;
        bool ok = tv && row < a.Cout;
        size_t idx;
        if (a.wtrans) { ok = ok && c < a.wO && row < a.wI; idx = ((size_t)tw * a.wO + c) * a.wI + row; }
        else { ok = ok && row < a.wO && c < a.wI; idx = ((size_t)tw * a.wO + row) * a.wI + c; }
        float x = ok ? a.w[idx] : 0.f;
        if (ok && a.wscale != nullptr) x *= a.wscale[row];     // eval: BatchNorm scale folded into the weights (as pack_all does)
        v[j] = x;
      }
      wf[b][s] = make_uint4(Tr<T>::pk(v[0], v[1]), Tr<T>::pk(v[2], v[3]), Tr<T>::pk(v[4], v[5]), Tr<T>::pk(v[6], v[7]));
    }
  }
  // lazy BN parameters of this thread's channel vector (staging)
  float sc[8], sh[8];
  const bool has_aff = a.scale != nullptr;
  {
    const int cv = tid % ThinCfg<CIN>::VPP;
#pragma unroll
    for (int i = 0; i < 8; ++i) { sc[i] = has_aff ? a.scale[cv * 8 + i] : 1.f; sh[i] = has_aff ? a.shift[cv * 8 + i] : 0.f; }
  }
  float bias[NB][4];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ch = b * 16 + 4 * g + j;
      bias[b][j] = (a.bias != nullptr && ch < a.Cout) ? a.bias[ch] : 0.f;
    }
  float s1[NB][4], s2[NB][4];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int j = 0; j < 4; ++j) { s1[b][j] = 0.f; s2[b][j] = 0.f; }
  const bool want_stats = a.slab != nullptr;
  const int ES = (int)sizeof(T);

  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int n = tile / (a.tiles_x * a.tiles_y);
    const int rem = tile - n * a.tiles_x * a.tiles_y;
    const int tyi = rem / a.tiles_x;
    const int y0 = tyi * TH, x0 = (rem - tyi * a.tiles_x) * THIN_TW;
    __syncthreads();                       // every wave is done reading the previous window
    thin_stage<T, CIN, TH>(a, win, n, y0, x0, tid, sc, sh, has_aff);
    __syncthreads();
    if (!a.pool) {
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr) {
        const int yy = wave * RPW + rr, y = y0 + yy;
#pragma unroll
        for (int seg = 0; seg < 2; ++seg) {
          const char* base = win + (yy * THIN_RW + seg * 16 + r16) * PIXB;
          f32x4_t acc[NB];
#pragma unroll
          for (int b = 0; b < NB; ++b) acc[b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < STEPS; ++s) {
            const uint4 bf = *(const uint4*)(base + toff[s]);
#pragma unroll
            for (int b = 0; b < NB; ++b) Tr<T>::mma16(wf[b][s], bf, acc[b]);
          }
          const int x = x0 + seg * 16 + r16;
          const bool pok = y < a.OH && x < a.OW;
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            float val[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              val[j] = acc[b][j] + bias[b][j];
              if (a.relu_out) val[j] = clamp_lo(val[j], 0.f);
              if (want_stats && pok) { s1[b][j] += val[j]; s2[b][j] += val[j] * val[j]; }
            }
            const int ch0 = b * 16 + 4 * g;
            if (!pok || ch0 >= a.Cout) continue;
            if (a.head) {
#pragma unroll
              for (int j = 0; j < 4; ++j)
                if (ch0 + j < a.Cout) ((float*)a.y)[(((size_t)n * a.Cout + ch0 + j) * a.yH + y) * a.yW + x] = val[j];
            } else {
              uint2* gp = (uint2*)(a.y + ((((size_t)n * a.yH + y) * a.yW + x) * a.yC + ch0) * ES);
              if (a.accum) {
                const uint2 old = *gp;
                val[0] += Tr<T>::lo(old.x); val[1] += Tr<T>::hi(old.x); val[2] += Tr<T>::lo(old.y); val[3] += Tr<T>::hi(old.y);
              }
              *gp = make_uint2(Tr<T>::pk(val[0], val[1]), Tr<T>::pk(val[2], val[3]));
            }
          }
        }
      }
    } else {
      // gradient of a nearest-x2 upsample: the 2x2 quad is summed in f32 -- rows 2q and 2q + 1 into ONE accumulator, the two
      // pixels of a lane pair by a cross-lane add -- and rounded once, at half resolution
#pragma unroll
      for (int rp = 0; rp < RPW / 2; ++rp) {
        const int yy = wave * RPW + 2 * rp, y = y0 + yy;
#pragma unroll
        for (int seg = 0; seg < 2; ++seg) {
          f32x4_t acc[NB];
#pragma unroll
          for (int b = 0; b < NB; ++b) acc[b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int dyr = 0; dyr < 2; ++dyr) {
            const char* base = win + ((yy + dyr) * THIN_RW + seg * 16 + r16) * PIXB;
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
              const uint4 bf = *(const uint4*)(base + toff[s]);
#pragma unroll
              for (int b = 0; b < NB; ++b) Tr<T>::mma16(wf[b][s], bf, acc[b]);
            }
          }
          const int x = x0 + seg * 16 + r16;
          const bool pok = y < a.OH && x < a.OW && (r16 & 1) == 0;      // (OH, OW even: a quad is inside the image as a whole)
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            float val[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) val[j] = acc[b][j] + __shfl_xor(acc[b][j], 1);
            const int ch0 = b * 16 + 4 * g;
            if (!pok || ch0 >= a.Cout) continue;
            uint2* gp = (uint2*)(a.y + ((((size_t)n * a.yH + (y >> 1)) * a.yW + (x >> 1)) * a.yC + ch0) * ES);
            if (a.accum) {
              const uint2 old = *gp;
              val[0] += Tr<T>::lo(old.x); val[1] += Tr<T>::hi(old.x); val[2] += Tr<T>::lo(old.y); val[3] += Tr<T>::hi(old.y);
            }
            *gp = make_uint2(Tr<T>::pk(val[0], val[1]), Tr<T>::pk(val[2], val[3]));
          }
        }
      }
    }
  }
  if (want_stats) {
    // fixed order: pixels of a lane, then the 16 lanes of a k-slice group, then the four waves -> deterministic
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float u = s1[b][j], v = s2[b][j];
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) { u += __shfl_xor(u, m); v += __shfl_xor(v, m); }
        if (r16 == 0) { red[((wave * NB * 16) + b * 16 + 4 * g + j) * 2] = u; red[((wave * NB * 16) + b * 16 + 4 * g + j) * 2 + 1] = v; }
      }
    __syncthreads();
    if (tid < NB * 16 && tid < a.Cout) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) { t1 += red[(w * NB * 16 + tid) * 2]; t2 += red[(w * NB * 16 + tid) * 2 + 1]; }
      float* o = a.slab + ((size_t)(a.slab_row0 + blockIdx.x) * a.Cout + tid) * 2;
      o[0] = t1; o[1] = t2;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------- weight gradient
typedef __attribute__((ext_vector_type(4))) short thin_s16x4_t;
typedef __attribute__((address_space(3))) thin_s16x4_t thin_lds_s16x4_t;

template <typename T, int CIN, int TH>
__global__ __launch_bounds__(THIN_NTHR, 2) void thin_wgrad_kernel(const ThinArgs a) {
  constexpr int PIXB = ThinCfg<CIN>::PIXB, NCB = CIN / 16;
  constexpr int RPW = TH / 4;
  constexpr int WINB = (TH + 2) * THIN_RW * PIXB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* win = smem;
  char* dyt = smem + WINB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;

  f32x4_t acc[9][NCB];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int b = 0; b < NCB; ++b) acc[t][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  int toff[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) toff[t] = (a.tdy[t] * THIN_RW + a.tdx[t]) * PIXB;
  // transposed-read addressing (ds_read_b64_tr_b16: per 16-lane group a block of 4 rows x 16 columns; lane 4q + p supplies row q,
  // columns 4p..4p+3 and receives column lane & 15 of the 4 rows).  k order of a 32-pixel row: group g reads pixels 4g..4g+3 (first
  // read) and 16+4g..16+4g+3 (second), the same for both operands; a 32-lane half then touches 8 consecutive pixels per read.
  const int ya0 = (4 * g + q) * 32 + p * 8;
  const int xa0 = (4 * g + q) * PIXB + p * 8;

  float sc[8], sh[8];
  const bool has_aff = a.scale != nullptr;
  {
    const int cv = tid % ThinCfg<CIN>::VPP;
#pragma unroll
    for (int i = 0; i < 8; ++i) { sc[i] = has_aff ? a.scale[cv * 8 + i] : 1.f; sh[i] = has_aff ? a.shift[cv * 8 + i] : 0.f; }
  }
  const int ES = (int)sizeof(T);
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int n = tile / (a.tiles_x * a.tiles_y);
    const int rem = tile - n * a.tiles_x * a.tiles_y;
    const int tyi = rem / a.tiles_x;
    const int y0 = tyi * TH, x0 = (rem - tyi * a.tiles_x) * THIN_TW;
    __syncthreads();
    thin_stage<T, CIN, TH>(a, win, n, y0, x0, tid, sc, sh, has_aff);
    {   // dy tile: TH x 32 pixels x 16 channels, two 16-byte vectors per pixel, zero outside the image
      constexpr int NV = TH * THIN_TW * 2, PER = NV / THIN_NTHR;
      uint4 v[PER];
      bool ok[PER];
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int vi = tid + u * THIN_NTHR, px = vi >> 1, cv = vi & 1;
        const int ty = px >> 5, tx = px & 31;
        const int y = y0 + ty, x = x0 + tx;
        ok[u] = y < a.OH && x < a.OW;
        const int yc = min(y, a.OH - 1), xc = min(x, a.OW - 1);
        v[u] = *(const uint4*)(a.dy + ((((size_t)n * a.OH + yc) * a.OW + xc) * a.dyC + cv * 8) * ES);
      }
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int vi = tid + u * THIN_NTHR;
        *(uint4*)(dyt + vi * 16) = ok[u] ? v[u] : make_uint4(0, 0, 0, 0);
      }
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
      const int yy = wave * RPW + rr;
      const char* yrow = dyt + yy * THIN_TW * 32 + ya0;
      struct Pair { thin_s16x4_t lo, hi; };
      Pair ya;
      ya.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((thin_lds_s16x4_t*)(yrow));
      ya.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((thin_lds_s16x4_t*)(yrow + 16 * 32));
      const uint4 yf = __builtin_bit_cast(uint4, ya);
      const char* xrow = win + yy * THIN_RW * PIXB + xa0;
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int b = 0; b < NCB; ++b) {
          Pair xb;
          xb.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((thin_lds_s16x4_t*)(xrow + toff[t] + b * 32));
          xb.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((thin_lds_s16x4_t*)(xrow + toff[t] + b * 32 + 16 * PIXB));
          Tr<T>::mma16(yf, __builtin_bit_cast(uint4, xb), acc[t][b]);
        }
    }
  }
  // ---- combine the four waves in LDS, then one round of fp32 atomics: D[row = co (4g + j)][col = ci (lane & 15)]
  __syncthreads();
  float* sum = (float*)smem;                                // [9][NCB][16 co][16 ci]
  constexpr int NSUM = 9 * NCB * 256;
  for (int i = tid; i < NSUM; i += THIN_NTHR) sum[i] = 0.f;
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int b = 0; b < NCB; ++b)
#pragma unroll
      for (int j = 0; j < 4; ++j) atomicAdd(&sum[((t * NCB + b) * 16 + 4 * g + j) * 16 + i16], acc[t][b][j]);
  __syncthreads();
  for (int i = tid; i < NSUM; i += THIN_NTHR) {
    const int ci = i & 15, co = (i >> 4) & 15, tb = i >> 8, b = tb % NCB, t = tb / NCB;
    const int cin = b * 16 + ci;
    if (co < a.Cout && cin < a.wI) atomicAdd(a.dW + ((size_t)a.tw[t] * a.wO + co) * a.wI + cin, sum[i]);
  }
}

// ---------------------------------------------------------------------------------------------------------------- host side
namespace {

static bool thin_on() {
  static const bool on = getenv("OCTSEG_NO_THIN") == nullptr;   // A/B switch
  return on;
}
static bool std33(const int* tdy, const int* tdx, int ntaps, int min_dy, int min_dx, int span_y, int span_x) {
  if (ntaps != 9 || span_y != 3 || span_x != 3 || min_dy != -1 || min_dx != -1) return false;
  unsigned seen = 0;
  for (int t = 0; t < 9; ++t) {
    const int oy = tdy[t] - min_dy, ox = tdx[t] - min_dx;
    if (oy < 0 || oy > 2 || ox < 0 || ox > 2) return false;
    seen |= 1u << (oy * 3 + ox);
  }
  return seen == 0x1ffu;
}
static int thin_th(int Cin) { return Cin == 16 ? 16 : 8; }
struct ThinGeom { int TH, tiles_x, tiles_y, ntiles, G; };
static ThinGeom thin_geom(int N, int OH, int OW, int Cin) {
  ThinGeom g;
  g.TH = thin_th(Cin);
  g.tiles_x = (OW + THIN_TW - 1) / THIN_TW; g.tiles_y = (OH + g.TH - 1) / g.TH;
  g.ntiles = N * g.tiles_x * g.tiles_y;
  g.G = g.ntiles < THIN_MAXWG ? g.ntiles : THIN_MAXWG;
  return g;
}

template <typename T, int CIN, int NB, int TH>
static hipError_t thin_launch_k(const ThinArgs& ta, int G, hipStream_t st) {
  const size_t lds = (size_t)(TH + 2) * THIN_RW * ThinCfg<CIN>::PIXB + (size_t)4 * NB * 16 * 2 * sizeof(float);
  hipLaunchKernelGGL((thin_conv_kernel<T, CIN, NB, TH>), dim3(G), dim3(THIN_NTHR), lds, st, ta);
  return hipGetLastError();
}
template <typename T>
static hipError_t thin_launch(const ThinArgs& ta, int Cin, int NB, int G, hipStream_t st) {
  if (Cin == 16) return NB == 1 ? thin_launch_k<T, 16, 1, 16>(ta, G, st) : thin_launch_k<T, 16, 2, 16>(ta, G, st);
  return NB == 1 ? thin_launch_k<T, 32, 1, 8>(ta, G, st) : thin_launch_k<T, 32, 2, 8>(ta, G, st);
}

}  // namespace

bool thin_conv_eligible(const ConvArgs& a, int dtype) {
  if (!thin_on() || dtype == DT_F32 || a.Wmaster == nullptr) return false;
  if (a.istride != 1 || a.ostride != 1 || a.ooy != 0 || a.oox != 0 || a.nsrc != 1 || a.ndst != 1) return false;
  if (!std33(a.tap_dy, a.tap_dx, a.ntaps, a.min_dy, a.min_dx, a.span_y, a.span_x)) return false;
  if (a.Cin != 16 && a.Cin != 32) return false;
  const SrcDesc& s = a.src[0];
  const DstDesc& d = a.dst[0];
  if (s.C != a.Cin || s.c0 != 0 || a.IH != a.OH || a.IW != a.OW || (s.H << s.up) != a.IH || (s.W << s.up) != a.IW) return false;
  if (a.out_mode == OUT_HEAD_NCHW) {
    if (a.Cout > 16 || d.pool) return false;
  } else {
    if ((a.Cout != 16 && a.Cout != 32) || d.c0 != 0 || d.cn != a.Cout || d.C != a.Cout) return false;
    if (d.pool ? (d.H * 2 != a.OH || d.W * 2 != a.OW) : (d.H != a.OH || d.W != a.OW)) return false;
  }
  if (a.bias != nullptr && a.stat_slab != nullptr) return false;
  // full-resolution maps only: small grids belong to the MFMA-bound kernels' occupancy rules
  return (long long)a.N * a.OH * a.OW >= 4096;
}

int thin_conv_rows(const ConvArgs& a) { return thin_geom(a.N, a.OH, a.OW, a.Cin).G; }

hipError_t launch_thin_conv(int dtype, const ConvArgs& a, hipStream_t st) {
  const ThinGeom g = thin_geom(a.N, a.OH, a.OW, a.Cin);
  ThinArgs ta;
  memset(&ta, 0, sizeof(ta));
  const SrcDesc& s = a.src[0];
  const DstDesc& d = a.dst[0];
  ta.x = (const char*)s.ptr; ta.scale = s.scale; ta.shift = s.shift; ta.relu = s.relu; ta.up = s.up; ta.sH = s.H; ta.sW = s.W;
  ta.N = a.N; ta.OH = a.OH; ta.OW = a.OW;
  for (int t = 0; t < 9; ++t) { ta.tdy[t] = a.tap_dy[t] - a.min_dy; ta.tdx[t] = a.tap_dx[t] - a.min_dx; ta.tw[t] = a.tap_w[t]; }
  ta.w = a.Wmaster; ta.wO = a.wO; ta.wI = a.wI; ta.wtrans = a.wtrans; ta.wscale = a.wscale;
  ta.Cout = a.Cout; ta.bias = a.bias; ta.relu_out = a.relu_out;
  ta.slab = a.stat_slab; ta.slab_row0 = a.slab_row0;
  ta.y = (char*)d.ptr; ta.yC = d.C; ta.yH = d.H; ta.yW = d.W; ta.accum = (d.accum || a.out_mode == OUT_ACCUM) ? 1 : 0; ta.pool = d.pool;
  ta.head = a.out_mode == OUT_HEAD_NCHW ? 1 : 0;
  ta.tiles_x = g.tiles_x; ta.tiles_y = g.tiles_y; ta.ntiles = g.ntiles;
  const int NB = a.Cout > 16 ? 2 : 1;
  if (dtype == DT_F16) return thin_launch<f16_t>(ta, a.Cin, NB, g.G, st);
  return thin_launch<bf16_t>(ta, a.Cin, NB, g.G, st);
}

bool thin_wgrad_eligible(const WgradArgs& a, int dtype) {
  if (!thin_on() || dtype != DT_BF16 || deterministic_mode()) return false;
  if (a.istride != 1 || a.dstride != 1 || a.doy != 0 || a.dox != 0 || a.nsrc != 1) return false;
  if (!std33(a.tap_dy, a.tap_dx, a.ntaps, a.min_dy, a.min_dx, a.span_y, a.span_x)) return false;
  if (a.Cin != 16 && a.Cin != 32) return false;
  const SrcDesc& s = a.src[0];
  if (s.C != a.Cin || s.c0 != 0 || a.IH != a.OH || a.IW != a.OW || (s.H << s.up) != a.IH || (s.W << s.up) != a.IW) return false;
  if (a.dyC != 16 || a.Cout > 16 || a.DH != a.OH || a.DW != a.OW) return false;
  return (long long)a.N * a.OH * a.OW >= 4096;
}

template <int CIN, int TH>
static hipError_t thin_wgrad_launch_k(const ThinArgs& ta, int G, hipStream_t st) {
  const size_t stage = (size_t)(TH + 2) * THIN_RW * ThinCfg<CIN>::PIXB + (size_t)TH * THIN_TW * 32;
  const size_t sum = (size_t)9 * (CIN / 16) * 256 * sizeof(float);
  hipLaunchKernelGGL((thin_wgrad_kernel<bf16_t, CIN, TH>), dim3(G), dim3(THIN_NTHR), stage > sum ? stage : sum, st, ta);
  return hipGetLastError();
}

hipError_t launch_thin_wgrad(int dtype, const WgradArgs& a, hipStream_t st) {
  (void)dtype;
  ThinGeom g = thin_geom(a.N, a.OH, a.OW, a.Cin);
  if (g.G > 512) g.G = 512;            // two workgroups per CU: fewer flush rounds, same streaming
  ThinArgs ta;
  memset(&ta, 0, sizeof(ta));
  const SrcDesc& s = a.src[0];
  ta.x = (const char*)s.ptr; ta.scale = s.scale; ta.shift = s.shift; ta.relu = s.relu; ta.up = s.up; ta.sH = s.H; ta.sW = s.W;
  ta.N = a.N; ta.OH = a.OH; ta.OW = a.OW;
  for (int t = 0; t < 9; ++t) { ta.tdy[t] = a.tap_dy[t] - a.min_dy; ta.tdx[t] = a.tap_dx[t] - a.min_dx; ta.tw[t] = a.tap_w[t]; }
  ta.wO = a.Cout; ta.wI = a.Cin; ta.Cout = a.Cout;
  ta.dy = (const char*)a.dy; ta.dyC = a.dyC; ta.dW = a.dW;
  ta.tiles_x = g.tiles_x; ta.tiles_y = g.tiles_y; ta.ntiles = g.ntiles;
  if (a.Cin == 16) return thin_wgrad_launch_k<16, 16>(ta, g.G, st);
  return thin_wgrad_launch_k<32, 8>(ta, g.G, st);
}

}  // namespace octseg
