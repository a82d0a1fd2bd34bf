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

#include <cstdlib>
#include <type_traits>

namespace octseg {

#ifdef OCTSEG_STAMP
#define WSTAMP(var) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define WSTAMP(var) do { } while (0)
#endif

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;

// MAXW = window passes held in registers by the prefetch pipeline (per tap count)
template <int NTAPS> struct WgradCfg { static constexpr int MAXW = NTAPS == 1 ? 4 : (NTAPS == 4 ? 6 : 7); };

// window-row-major contraction (pipelined == 2): MFMA slots of a tile before window row wr -- row w carries KS MFMAs for every output row
// kk in [w - KS + 1, w] that exists
__host__ __device__ constexpr int wrow_slots_before(int wr, int ks, int th) {
  int n = 0;
  for (int w = 0; w < wr; ++w) {
    const int klo = w - (ks - 1) < 0 ? 0 : w - (ks - 1), khi = w < th ? w : th - 1;
    n += (khi - klo + 1) * ks;
  }
  return n;
}

template <typename T, int NTAPS>
__global__ __launch_bounds__(NTHR) void wgrad_mfma_kernel(const WgradArgs a, const int th, const int pipelined) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int VEC = Tr<T>::VEC;
  constexpr int RB = 64 * (int)sizeof(T);  // 64 channels per LDS row
  // bf16: the transposed reads of a 32-lane group touch 4 pixel rows x 16 banks each: the row pitch must move the
  // bank by 16 (mod 64) per row -> 192 bytes (144 makes rows 0/2 and 1/3 collide: every read 2-way conflicted)
  constexpr int PITCH = sizeof(T) == 2 ? 192 : RB + 16;
  constexpr int VPR = RB / 16;
  constexpr int PSTEP = NTHR / VPR;
  constexpr int MAXY = 4, MAXW = WgradCfg<NTAPS>::MAXW;
  typedef WindowStager<T, RB, NTHR, PITCH> Stager;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wq_m = wave >> 1, wq_n = wave & 1;
  // XCD-aware placement.  Workgroups are dealt round-robin over the 8 XCDs (linear id % 8), each with its own L2.  All
  // (ci, co) tiles of one pixel slice read the same dy and x tiles, so a pixel slice is kept on ONE XCD (its operands
  // are then fetched into one L2 instead of eight: the TCC counters showed ~3x the algorithmic bytes).  Speed only.
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  {
    const int gx_ = gridDim.x, gy_ = gridDim.y, gz_ = gridDim.z;
    const int total = gx_ * gy_ * gz_;
    // Round 4: the placement is made of GROUPS -- the gy co-tile workgroups of one (ci tile, pixel slice) share the x tile -- dealt to the XCDs
    // whole, consecutive groups of a pixel slice to the same XCD round (they share the dy tiles).  It needs gx * gz groups divisible by 8;
    // round 3's form (whole pixel slices per XCD) needed gz % 8 == 0 and was therefore OFF on the big decoder layers, where split-K is 4:
    // 1024->256 @176^2 fetched 3.05 GB for 1.27 GB of operands.  a.co_fast = 0 (OCTSEG_WGRAD_CI_MAJOR=1) restores round 3's rule for A/B.
    const int groups = gx_ * gz_;
    const int lid = bx + gx_ * (by + gy_ * bz);
    if (a.co_fast && (groups & 7) == 0) {
      const int x = lid & 7, q = lid >> 3;          // XCD, position inside the XCD's share
      const int k = q / gy_, r = q - k * gy_;
      const int g = x * (groups >> 3) + k;          // XCD x owns groups [x G / 8, (x + 1) G / 8): neighbouring ci tiles of a pixel slice share dy in ONE L2
      by = r; bz = g / gx_; bx = g - bz * gx_;
    } else if (!a.co_fast && (total & 7) == 0 && (gz_ & 7) == 0) {
      const int w = (lid & 7) * (total >> 3) + (lid >> 3);   // XCD k owns the z range [k * gz / 8, (k + 1) * gz / 8)
      bz = w / (gx_ * gy_);
      const int rem = w - bz * (gx_ * gy_);
      by = rem / gx_; bx = rem - by * gx_;
    }
  }
  const int ci0 = bx * 64, co0 = by * 64;

  const int tiles_x = (a.OW + TW - 1) / TW, tiles_y = (a.OH + th - 1) / th;
  const int ntiles = a.N * tiles_x * tiles_y;

  const bool single = a.ntaps == 1;
  const int lstride = single ? 1 : a.istride;
  const int smul = single ? a.istride : 1;
  const int RH = single ? th : (th - 1) * a.istride + a.span_y;
  const int RW = single ? TW : (TW - 1) * a.istride + a.span_x;
  const int npix = RH * RW;
  const int npw = (npix + PSTEP - 1) / PSTEP;      // window passes
  const int npy = (th * TW + PSTEP - 1) / PSTEP;   // dy-tile passes
  const float inv_rw = 1.0f / (float)RW;

  char* ldsY = smem;                          // [npy*PSTEP][64 ch] dy tile (rows padded to whole passes)
  char* ldsX = smem + npy * PSTEP * PITCH;    // [npw*PSTEP][64 ch] input window

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

  Stager sg;  // window stager: the channel chunk is fixed for the whole kernel
  sg.setup(a.src, a.nsrc, a.Cin, bx, tid);
  const int ycv = tid % VPR, yp0 = tid / VPR;
  const int yc = co0 + ycv * VEC;
  const bool ycok = yc < a.dyC;

  struct TilePos { int n, y0, x0; };
  auto tile_pos = [&](int tile) {
    TilePos tp;
    tp.n = tile / (tiles_x * tiles_y);
    const int rem = tile - tp.n * tiles_x * tiles_y;
    const int tyi = rem / tiles_x;
    tp.y0 = tyi * th; tp.x0 = (rem - tyi * tiles_x) * TW;
    return tp;
  };
  // branch-free dy load: clamped address, validity resolved when the vector is written to LDS.  All per-pass
  // coordinates are tile invariant and precomputed; per tile only a 32-bit offset from the image base is formed.
  const int dy_pix_bytes = a.dyC * (int)sizeof(T), dy_row_bytes = a.DW * dy_pix_bytes;
  auto load_y = [&](const char* ybase, const TilePos& tp, int ty, int tx, bool in_tile, bool& ok) {
    const int gy = tp.y0 + ty, gx = tp.x0 + tx;
    ok = ycok && in_tile && gy < a.OH && gx < a.OW;
    const int gyc = min(gy, a.OH - 1), gxc = min(gx, a.OW - 1);
    const unsigned off = (unsigned)((gyc * a.dstride + a.doy) * dy_row_bytes + (gxc * a.dstride + a.dox) * dy_pix_bytes);
    return *(const uint4*)(ybase + off);
  };
  auto write_y = [&](int pass, uint4 v, bool ok) {
    *(uint4*)(ldsY + (pass * PSTEP + yp0) * PITCH + ycv * 16) = ok ? v : make_uint4(0, 0, 0, 0);
  };
  auto y_base = [&](int n) { return (const char*)a.dy + ((size_t)n * a.DH * a.DW * a.dyC + (ycok ? yc : 0)) * sizeof(T); };
  auto compute = [&]() {
    if constexpr (sizeof(T) == 2) {
      // One wave per SIMD (the NTAPS accumulators leave no room for a second): nobody else hides this wave's LDS
      // latency, so the transposed fragment reads of pixel row kk+1 are issued before the MFMAs of row kk
      // (two register sets; the spare half of the 512-register file pays for them).
      struct Pair { s16x4_t lo, hi; };  // two transposed 4-element reads = one 8-element MFMA fragment
      struct Frags { uint4 y; uint4 x[NTAPS]; };
      auto load_frags = [&](int kk, Frags& f) {
        const char* yrow = ldsY + kk * TW * PITCH + ya0;
        const char* xrow = ldsX + (kk * lstride) * RW * PITCH + xa0;
        Pair ya;
        ya.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(yrow));
        ya.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(yrow + 4 * PITCH));
        f.y = __builtin_bit_cast(uint4, ya);
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) {
          Pair xb;
          xb.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(xrow + toff[t]));
          xb.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(xrow + toff[t] + xrow_step));
          f.x[t] = __builtin_bit_cast(uint4, xb);
        }
      };
      auto load_y1 = [&](int kk, Frags& f) {
        const char* yrow = ldsY + kk * TW * PITCH + ya0;
        Pair ya;
        ya.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(yrow));
        ya.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(yrow + 4 * PITCH));
        f.y = __builtin_bit_cast(uint4, ya);
      };
      auto load_x1 = [&](const char* xrow, int t, Frags& f) {
        Pair xb;
        xb.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(xrow + toff[t]));
        xb.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(xrow + toff[t] + xrow_step));
        f.x[t] = __builtin_bit_cast(uint4, xb);
      };
      // MFMAs of row `cur`, the fragment reads of row kn spread between them (two reads per MFMA gap)
      auto row_step = [&](const Frags& cur, int kn, Frags& nxt) {
        const char* xrow = ldsX + (kn * lstride) * RW * PITCH + xa0;
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) {
          Tr<T>::mma(cur.y, cur.x[t], acc[t]);
          __builtin_amdgcn_sched_barrier(0);
          if (t == 0) load_y1(kn, nxt);
          load_x1(xrow, t, nxt);
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      Frags fa, fb;
      load_frags(0, fa);
      for (int kk = 0; kk < th; kk += 2) {       // th is even on this path (2, 4 or 8); rows past th - 1 are re-reads
        row_step(fa, min(kk + 1, th - 1), fb);
        if (kk + 1 < th) row_step(fb, min(kk + 2, th - 1), fa);
      }
    } else {
      for (int kk = 0; kk < th; ++kk) {
        const char* yrow = ldsY + kk * TW * PITCH + ya0;
        const char* xrow = ldsX + (kk * lstride) * RW * PITCH + xa0;
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
  };

#ifdef OCTSEG_STAMP
  unsigned long long wts[5] = {0, 0, 0, 0, 0};
#endif
  if (pipelined) {
    // register-prefetch pipeline: the global loads of tile k+1 fly under the MFMAs of tile k; every
    // load / LDS store is unconditional (clamped indices) so that no wait hides behind a branch.
    uint4 yv[MAXY], xv[MAXW];
    bool yok[MAXY], xok[MAXW];
    int ypos[MAXY], wpos[MAXW];     // packed (row << 16 | col) of every pass, tile invariant
    bool yin[MAXY], win[MAXW];
#pragma unroll
    for (int u = 0; u < MAXY; ++u) {
      const int p = min(u, npy - 1) * PSTEP + yp0;
      ypos[u] = ((p >> 4) << 16) | (p & 15);
      yin[u] = p < th * TW;
    }
#pragma unroll
    for (int u = 0; u < MAXW; ++u) {
      const int hp = min(u, npw - 1) * PSTEP + sg.p0;
      const int hy = (int)(((float)hp + 0.5f) * inv_rw);
      wpos[u] = (hy << 16) | (hp - hy * RW);
      win[u] = hp < npix;
    }
    auto load_tile = [&](int tile) {
      const TilePos tp = tile_pos(tile);
      const char* yb = y_base(tp.n);
      sg.bind_image(tp.n);
      const int gy0 = tp.y0 * a.istride + a.min_dy, gx0 = tp.x0 * a.istride + a.min_dx;
#pragma unroll
      for (int u = 0; u < MAXY; ++u) yv[u] = load_y(yb, tp, ypos[u] >> 16, ypos[u] & 0xffff, yin[u], yok[u]);
#pragma unroll
      for (int u = 0; u < MAXW; ++u)
        xv[u] = sg.load_at(wpos[u] >> 16, wpos[u] & 0xffff, win[u], gy0, gx0, smul, a.IH, a.IW, xok[u]);
    };
    auto write_tile = [&]() {
#pragma unroll
      for (int u = 0; u < MAXY; ++u) write_y(min(u, npy - 1), yv[u], yok[u]);
#pragma unroll
      for (int u = 0; u < MAXW; ++u) sg.write(ldsX, min(u, npw - 1), xv[u], xok[u]);
    };
    int tile = bz;
    if constexpr (sizeof(T) == 2 && (NTAPS == 9 || NTAPS == 4)) {
      if (pipelined == 2 && tile < ntiles) {
        // Two LDS tile buffers, ONE barrier per tile, and nothing but MFMAs on the critical path: while the 8 pixel rows
        // of tile k are contracted out of buffer k&1, the register-held loads of tile k+1 are stored into the other
        // buffer (rows 0-3) and the global loads of tile k+2 are issued into the same registers (rows 4-7), all as
        // fillers in the MFMA gaps (the wave is alone on its SIMD: a separate issue / store phase idles the pipe).
        // KS = kernel extent (3x3, or the 2x2 taps of one ConvTranspose2d parity -- host-checked tap order, launch_wgrad_t): window rows of
        // 16 + KS - 1 pixels, TH8 + KS - 1 of them
        constexpr int TH8 = 8, NSLOT = TH8 * NTAPS, NITEM = MAXY + MAXW, KS = NTAPS == 9 ? 3 : 2, RWC = TW + KS - 1, NWR = TH8 + KS - 1;
        const int tile_bytes = (npy + npw) * PSTEP * PITCH;
        TilePos tp2{0, 0, 0};
        const char* yb2 = nullptr;
        int gy02 = 0, gx02 = 0;
        auto item = [&](auto jc, auto phase, char* oy, char* ox) __attribute__((always_inline)) {
          constexpr int J = decltype(jc)::value;
          constexpr int PH = decltype(phase)::value;
          if constexpr (PH == 0) {            // store pass J of the tile held in registers
            if constexpr (J < MAXY) *(uint4*)(oy + (min(J, npy - 1) * PSTEP + yp0) * PITCH + ycv * 16) = yok[J] ? yv[J] : make_uint4(0, 0, 0, 0);
            else sg.write(ox, min(J - MAXY, npw - 1), xv[J - MAXY], xok[J - MAXY]);
          } else {                            // load pass J of the tile after next
            if constexpr (J < MAXY) yv[J] = load_y(yb2, tp2, ypos[J] >> 16, ypos[J] & 0xffff, yin[J], yok[J]);
            else xv[J - MAXY] = sg.load_at(wpos[J - MAXY] >> 16, wpos[J - MAXY] & 0xffff, win[J - MAXY], gy02, gx02, smul, a.IH, a.IW, xok[J - MAXY]);
          }
        };
        // filler slot S of the tile (one per MFMA): the items whose share of the half tile falls on it
        auto fillers = [&](auto sc_, char* oy, char* ox) __attribute__((always_inline)) {
          constexpr int S = decltype(sc_)::value;
          constexpr int HALF = NSLOT / 2;
          constexpr int PH = S < HALF ? 0 : 1;
          constexpr int SS = S - PH * HALF;
          constexpr int J0 = (SS * NITEM + HALF - 1) / HALF, J1 = ((SS + 1) * NITEM + HALF - 1) / HALF;   // ceil ranges
          if constexpr (J0 < J1 && J0 < NITEM) item(std::integral_constant<int, J0>{}, std::integral_constant<int, PH>{}, oy, ox);
          if constexpr (J0 + 1 < J1 && J0 + 1 < NITEM) item(std::integral_constant<int, J0 + 1>{}, std::integral_constant<int, PH>{}, oy, ox);
          if constexpr (J0 + 2 < J1 && J0 + 2 < NITEM) item(std::integral_constant<int, J0 + 2>{}, std::integral_constant<int, PH>{}, oy, ox);
        };
        // WINDOW-ROW-major contraction (round 4).  Tap (r, s) of output row kk reads window row kk + r at column shift s, so the x fragment
        // of (window row w, shift s) serves tap (0, s) of row w, tap (1, s) of row w - 1 and tap (2, s) of row w - 2: read ONCE per tile
        // and used for three MFMAs (against three rotating dy fragments) instead of being re-read per tap -- 30 x + 8 dy fragments per
        // tile instead of 72 + 8, 1.05 transposed reads per MFMA instead of 2.2 (counters of round 3's loop, profiles/
        // r4_sq_counters_wgrad9_*.txt: 2.4 LDS + 4 vector + 1.3 scalar instructions per MFMA on a wave that is alone on its SIMD, 48 % of
        // the cycles in the MFMA pipe).  Window row w carries 3 / 6 / 9 MFMAs (rows 0 and 9 / 1 and 8 / 2..7): 72 per tile as before,
        // the same slot numbering for the staging fillers, fragments of window row w + 1 requested in the gaps of row w.
        struct Frag { s16x4_t lo, hi; };
        Frag xs[2][KS], ys[4];
        auto rd_x = [&](const char* xbase, auto wc, auto sc, auto hc) __attribute__((always_inline)) {
          constexpr int Wn = decltype(wc)::value, S_ = decltype(sc)::value, H_ = decltype(hc)::value;
          constexpr int OFF = (Wn * RWC + S_) * PITCH + H_ * 4 * PITCH;
          if constexpr (H_ == 0) xs[Wn & 1][S_].lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(xbase + OFF));
          else xs[Wn & 1][S_].hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(xbase + OFF));
        };
        auto rd_y = [&](const char* ybase, auto kc, auto hc) __attribute__((always_inline)) {
          constexpr int Kn = decltype(kc)::value, H_ = decltype(hc)::value;
          constexpr int OFF = Kn * TW * PITCH + H_ * 4 * PITCH;
          if constexpr (H_ == 0) ys[Kn & 3].lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(ybase + OFF));
          else ys[Kn & 3].hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(ybase + OFF));
        };
        // read q of the prefetch list of window row Wn: x (shift 0 .. KS - 1) x (lo, hi), then dy (lo, hi)
        auto prefetch = [&](auto wc, auto qc, const char* xbase, const char* ybase) __attribute__((always_inline)) {
          constexpr int Q = decltype(qc)::value;
          if constexpr (Q < 2 * KS) rd_x(xbase, wc, std::integral_constant<int, Q / 2>{}, std::integral_constant<int, Q % 2>{});
          else rd_y(ybase, wc, std::integral_constant<int, Q - 2 * KS>{});
        };
        auto wrow = [&](auto wc, const char* by, const char* bx, char* oy, char* ox) __attribute__((always_inline)) {
          constexpr int Wr = decltype(wc)::value;                     // window row 0 .. NWR - 1
          constexpr int KLO = Wr - (KS - 1) < 0 ? 0 : Wr - (KS - 1), KHI = Wr < TH8 ? Wr : TH8 - 1;
          constexpr int CNT = (KHI - KLO + 1) * KS;                   // MFMAs of this window row
          constexpr int S0 = wrow_slots_before(Wr, KS, TH8);          // slots before it (3x3: 3, 6, then 9 per row, 6, 3)
          constexpr int NR = Wr + 1 >= NWR ? 0 : (Wr + 1 < TH8 ? 2 * KS + 2 : 2 * KS);   // reads for window row Wr + 1 (dy rows end at 7)
          const char* xbase = bx + xa0;
          const char* ybase = by + ya0;
          auto one = [&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            if constexpr (j < CNT) {
              // accumulator of (window row - output row, column shift): 3x3 taps ascend with the window offset, a ConvT parity's descend
              constexpr int kk = KLO + j / KS, sft = j % KS, t = NTAPS == 9 ? (Wr - kk) * 3 + sft : (1 - (Wr - kk)) * 2 + (1 - sft);
              Tr<T>::mma(__builtin_bit_cast(uint4, ys[kk & 3]), __builtin_bit_cast(uint4, xs[Wr & 1][sft]), acc[t]);
              __builtin_amdgcn_sched_barrier(0);
              constexpr int R0 = j * NR / CNT, R1 = (j + 1) * NR / CNT;
              if constexpr (R0 < R1) prefetch(std::integral_constant<int, Wr + 1>{}, std::integral_constant<int, R0>{}, xbase, ybase);
              if constexpr (R0 + 1 < R1) prefetch(std::integral_constant<int, Wr + 1>{}, std::integral_constant<int, R0 + 1>{}, xbase, ybase);
              if constexpr (R0 + 2 < R1) prefetch(std::integral_constant<int, Wr + 1>{}, std::integral_constant<int, R0 + 2>{}, xbase, ybase);
              fillers(std::integral_constant<int, S0 + j>{}, oy, ox);
              __builtin_amdgcn_sched_barrier(0);
            }
          };
          one(std::integral_constant<int, 0>{}); one(std::integral_constant<int, 1>{}); one(std::integral_constant<int, 2>{});
          one(std::integral_constant<int, 3>{}); one(std::integral_constant<int, 4>{}); one(std::integral_constant<int, 5>{});
          one(std::integral_constant<int, 6>{}); one(std::integral_constant<int, 7>{}); one(std::integral_constant<int, 8>{});
        };
        // prologue: tile 0 into buffer 0, tile 1 into the registers
        load_tile(tile);
        write_tile();
        load_tile(tile + a.ksplit < ntiles ? tile + a.ksplit : tile);
        __syncthreads();
        int cur = 0;
        for (; tile < ntiles; tile += a.ksplit) {
          const int t2 = tile + 2 * a.ksplit < ntiles ? tile + 2 * a.ksplit : tile;   // clamped: a re-fetch at the tail is harmless
          tp2 = tile_pos(t2);
          yb2 = y_base(tp2.n);
          gy02 = tp2.y0 * a.istride + a.min_dy; gx02 = tp2.x0 * a.istride + a.min_dx;
          const char* by = smem + cur * tile_bytes;
          const char* bx = by + npy * PSTEP * PITCH;
          char* oy = smem + (cur ^ 1) * tile_bytes;
          char* ox = oy + npy * PSTEP * PITCH;
          {
            auto z = std::integral_constant<int, 0>{};
            auto o = std::integral_constant<int, 1>{};
            rd_y(by + ya0, z, z); rd_y(by + ya0, z, o);
            rd_x(bx + xa0, z, z, z); rd_x(bx + xa0, z, z, o);
            rd_x(bx + xa0, z, o, z); rd_x(bx + xa0, z, o, o);
            if constexpr (KS == 3) { rd_x(bx + xa0, z, std::integral_constant<int, 2>{}, z); rd_x(bx + xa0, z, std::integral_constant<int, 2>{}, o); }
          }
          // loads of tile k+2 address the image through sg: bind it before the first load item, which sits in the second half of the tile
          // (3x3: slot 36 = the first MFMA of window row 5; 2x2: slot 16, inside window row 4) -- the stores of tile k+1 in the first
          // half only need scale / shift
          wrow(std::integral_constant<int, 0>{}, by, bx, oy, ox);
          wrow(std::integral_constant<int, 1>{}, by, bx, oy, ox);
          wrow(std::integral_constant<int, 2>{}, by, bx, oy, ox);
          wrow(std::integral_constant<int, 3>{}, by, bx, oy, ox);
          if constexpr (KS == 2) sg.bind_image(tp2.n);
          wrow(std::integral_constant<int, 4>{}, by, bx, oy, ox);
          if constexpr (KS == 3) sg.bind_image(tp2.n);
          wrow(std::integral_constant<int, 5>{}, by, bx, oy, ox);
          wrow(std::integral_constant<int, 6>{}, by, bx, oy, ox);
          wrow(std::integral_constant<int, 7>{}, by, bx, oy, ox);
          wrow(std::integral_constant<int, 8>{}, by, bx, oy, ox);
          if constexpr (NWR > 9) wrow(std::integral_constant<int, 9>{}, by, bx, oy, ox);
          __builtin_amdgcn_sched_barrier(0);
          __syncthreads();
          cur ^= 1;
        }
        tile = ntiles;   // done: skip the single-buffer loop below
      }
    }
    if (tile < ntiles) {
      load_tile(tile);
      write_tile();
      __syncthreads();
      for (; tile < ntiles; tile += a.ksplit) {
        unsigned long long w0 = 0, w1 = 0, w2 = 0, w3 = 0, w4 = 0; (void)w0; (void)w1; (void)w2; (void)w3; (void)w4;
        WSTAMP(w0);
        const int nxt = tile + a.ksplit < ntiles ? tile + a.ksplit : tile;  // last round re-fetches (harmless)
        load_tile(nxt);
        WSTAMP(w1);
        compute();
        WSTAMP(w2);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();  // every wave is done reading the tile
        WSTAMP(w3);
        write_tile();
        __syncthreads();
        WSTAMP(w4);
#ifdef OCTSEG_STAMP
        wts[0] += w1 - w0; wts[1] += w2 - w1; wts[2] += w3 - w2; wts[3] += w4 - w3; wts[4] += 1;
#endif
      }
    }
  } else {
    for (int tile = bz; tile < ntiles; tile += a.ksplit) {
      const TilePos tp = tile_pos(tile);
      __syncthreads();  // previous tile fully consumed
      for (int p = 0; p < npy; p += MAXY) {
        uint4 v[MAXY]; bool ok[MAXY];
#pragma unroll
        for (int u = 0; u < MAXY; ++u) {
          const int pp = min(p + u, npy - 1) * PSTEP + yp0;
          v[u] = load_y(y_base(tp.n), tp, pp >> 4, pp & 15, pp < th * TW, ok[u]);
        }
#pragma unroll
        for (int u = 0; u < MAXY; ++u) write_y(min(p + u, npy - 1), v[u], ok[u]);
      }
      for (int p = 0; p < npw; p += 4) {
        uint4 v[4]; bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          v[u] = sg.load(min(p + u, npw - 1), tp.n, tp.y0 * a.istride + a.min_dy, tp.x0 * a.istride + a.min_dx, smul, RW,
                         npix, inv_rw, a.IH, a.IW, ok[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) sg.write(ldsX, min(p + u, npw - 1), v[u], ok[u]);
      }
      __syncthreads();
      compute();
    }
  }

#ifdef OCTSEG_STAMP
  if (a.stamp != nullptr && lane == 0)
    for (int i = 0; i < 5; ++i) atomicAdd(a.stamp + i, wts[i]);
#endif
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

struct WgradGeom { int npy, npw; size_t lds; };
static WgradGeom wgrad_geom(const WgradArgs& a, int dtype, int th) {
  const int RB = 64 * (int)dtype_size(dtype), PITCH = dtype_size(dtype) == 2 ? 192 : RB + 16, PSTEP = NTHR / (RB / 16);
  const bool single = a.ntaps == 1;
  const int RH = single ? th : (th - 1) * a.istride + a.span_y;
  const int RW = single ? TW : (TW - 1) * a.istride + a.span_x;
  WgradGeom g;
  g.npy = (th * TW + PSTEP - 1) / PSTEP;
  g.npw = (RH * RW + PSTEP - 1) / PSTEP;
  g.lds = (size_t)(g.npy + g.npw) * PSTEP * PITCH;
  return g;
}

template <typename T, int NTAPS>
static hipError_t launch_wgrad_t(const WgradArgs& a0, int dtype, hipStream_t st) {
  WgradArgs a = a0;
  // largest pixel tile whose passes fit the register pipeline (MAXY / MAXW); else largest that fits LDS
  int th = 0, pipelined = 0;
  for (int cand = 8; cand >= 2; cand >>= 1) {
    const WgradGeom g = wgrad_geom(a, dtype, cand);
    if (g.npy <= 4 && g.npw <= WgradCfg<NTAPS>::MAXW && g.lds <= 150 * 1024) { th = cand; pipelined = 1; break; }
  }
  if (!th) {
    th = 8;
    while (th > 1 && wgrad_geom(a, dtype, th).lds > 150 * 1024) th >>= 1;
  }
  size_t lds = wgrad_geom(a, dtype, th).lds;
  // bf16, 8-row tiles, 3x3: double-buffered tiles with the stores / loads as MFMA fillers (one barrier per tile)
  static const bool no_pipe2 = getenv("OCTSEG_NO_WGRAD_PIPE2") != nullptr;
  bool std33 = NTAPS == 9 && a.istride == 1 && a.span_x == 3 && a.span_y == 3;
  for (int t = 0; std33 && t < 9; ++t) std33 = a.tap_dy[t] - a.min_dy == t / 3 && a.tap_dx[t] - a.min_dx == t % 3;
  // ... and the 2x2 taps of one ConvTranspose2d parity (wgrad_launches' order: both offsets descending) over dy's parity plane
  static const bool no_pipe2_4 = getenv("OCTSEG_NO_WGRAD_PIPE2_4") != nullptr;   // A/B switch
  bool std22 = NTAPS == 4 && !no_pipe2_4 && a.istride == 1 && a.span_x == 2 && a.span_y == 2;
  for (int t = 0; std22 && t < 4; ++t) std22 = a.tap_dy[t] - a.min_dy == 1 - t / 2 && a.tap_dx[t] - a.min_dx == 1 - t % 2;
  if (!no_pipe2 && pipelined && th == 8 && dtype != DT_F32 && (std33 || std22) && 2 * lds <= 150 * 1024) { pipelined = 2; lds *= 2; }
  const int ntiles = a.N * ((a.OW + TW - 1) / TW) * ((a.OH + th - 1) / th);
  const int gx = (a.Cin + 63) / 64, gy = (a.Cout + 63) / 64;
  // One resident round at most (two workgroup slots per CU: 516 workgroups take twice as long as 504).  Every workgroup ends by
  // adding its 64 x 64 x taps fp32 tile to dW with atomics: 504 x 147 KiB = 74 MB of atomic traffic per 3x3 layer, which is what
  // bounds the encoder's 3x3 weight gradients (small maps, 256..512 channels) -- half the workgroups (one per CU, twice the
  // pixels each) halve it: encoder 3x3 wgrad 3.89 -> 2.98 ms per step, decoder 14.7 -> 14.4.  1x1 layers flush 16 KiB tiles and want
  // the occupancy: 512 there (256 measured 2.7 -> 3.7 ms).
  static const int wg_env = getenv("OCTSEG_WGRAD_WGS") ? atoi(getenv("OCTSEG_WGRAD_WGS")) : 0;   // experiments
  const int wg_target = wg_env > 0 ? wg_env : (NTAPS == 1 ? 512 : (a.wg_target > 0 ? a.wg_target : 256));
  int ks = wg_target / (gx * gy);
  if (ks > ntiles) ks = ntiles;
  if (ks < 1) ks = 1;
  if (deterministic_mode()) ks = 1;   // every dW element then has ONE writer (its atomicAdd meets a zeroed buffer): a fixed summation order
  a.ksplit = ks;
  static const bool ci_major = getenv("OCTSEG_WGRAD_CI_MAJOR") != nullptr;   // A/B switch
  a.co_fast = ci_major ? 0 : 1;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)wgrad_mfma_kernel<T, NTAPS>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL((wgrad_mfma_kernel<T, NTAPS>), dim3(gx, gy, ks), dim3(NTHR), lds, st, a, th, pipelined);
  return hipGetLastError();
}

// Tap tables with other sizes are split host-side into groups the instantiations cover.
// stride-1 1x1: the pixel list is geometry free -> one image of 16-pixel rows, every tile full
static void flatten_1x1(WgradArgs& a) {
  if (a.ntaps != 1 || a.istride != 1 || a.dstride != 1 || a.tap_dy[0] != 0 || a.tap_dx[0] != 0) return;
  const long long npix = (long long)a.N * a.OH * a.OW;
  if (npix % TW != 0 || a.IH != a.OH || a.IW != a.OW || a.DH != a.OH || a.DW != a.OW) return;
  for (int i = 0; i < a.nsrc; ++i)
    if (a.src[i].up || a.src[i].H != a.IH || a.src[i].W != a.IW) return;
  const int rows = (int)(npix / TW);
  a.N = 1; a.IH = a.OH = a.DH = rows; a.IW = a.OW = a.DW = TW;
  for (int i = 0; i < a.nsrc; ++i) { a.src[i].H = rows; a.src[i].W = TW; }
}

hipError_t launch_wgrad(int dtype, const WgradArgs& a0, hipStream_t st) {
  if (dtype == DT_F16) return hipErrorInvalidValue;   // f16 is an eval-forward dtype

  if (a0.ntaps <= 0) return hipSuccess;
  if (thin_wgrad_eligible(a0, dtype)) return launch_thin_wgrad(dtype, a0, st);
  WgradArgs a = a0;
  flatten_1x1(a);
  if (wgrad1x1_eligible(a, dtype)) return launch_wgrad1x1(dtype, a, st);
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
