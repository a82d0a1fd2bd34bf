// wgrad_convt.hip -- weight gradient of ConvTranspose2d(k4, s2, p1) on gfx950 matrix cores, all 16 taps in ONE launch (bf16).
//
//   dW[r][s][co][ci] = sum over the low-resolution pixels (y, x) of  dy[2y + py][2x + px][co] * xin[y + d(py, r)][x + d(px, s)][ci],
//   d(p, k) = (p + 1 - k) / 2 for the k with p + 1 - k even: every output parity (py, px) meets its own 2 x 2 of the 16 taps.
//
// wgrad_mfma.hip runs this as four launches (one parity each): each stages the SAME input window again, and a 4-tap tile of 64 x 64 channels
// only contracts 4.2 MFLOP out of its 36 KB of operands -- the launches are bound by operand traffic (window-row-major 4-tap loop:
// 580-650 TFLOP/s where the 9-tap loop holds 950, profiles/r4_tied_probe3.txt).  Here a workgroup keeps the 10 x 18 window of an 8 x 16 tile
// of low-resolution pixels in LDS while dy's four parity planes of that tile stream through: 87 KB of operands per 16.8 MFLOP instead of 142.
//   * LDS: two window buffers (tile k / k + 1) and two dy-plane buffers (plane q / q + 1), pixel-major as in HBM, transposed reads
//     (ds_read_b64_tr_b16) straight into the MFMA operands -- the layout, row pitch and lane map of wgrad_mfma.hip.
//   * one barrier per PLANE (32 MFMAs per wave); in their gaps: the four LDS stores of the next plane (first half), the four global loads
//     of the plane after next (second half), and three of the twelve window stores / loads that move tile k + 1 into the other window buffer
//     and tile k + 2 into registers (stores in planes 0-1, loads in planes 2-3).  Eleven fillers per 32 MFMAs where the per-parity launch
//     needs twenty.
//   * contraction window-row-major: the x fragment of (window row w, column shift c) serves the taps of output rows w - py and w - py - 1.
//   * 16 accumulators of 32 x 32 per wave (256 registers: the wave is alone on its SIMD), flushed with fp32 atomics as 128-byte row segments.
// The input window is the forward's virtual tensor (lazy BN + ReLU applied while staging).  Used for LinkNet's decoder (smp DecoderBlock's
// ConvTranspose2d, reference sweep configs/tune.yaml:9-18) and for the tied decomposition of the nearest-x2 + 3x3 decoder layers
// (plan.h ConvLayer::tie), whose 4x4 gradient this is.
#include "common.h"
#include "conv_common.h"
#include "kernels.h"

#include <cstdlib>
#include <type_traits>

namespace octseg {

namespace {
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;

constexpr int CT_TH = 8, CT_RW = TW + 2, CT_RH = CT_TH + 2, CT_NPIX = CT_RW * CT_RH;
constexpr int CT_PITCH = 192, CT_VPR = 8, CT_PSTEP = NTHR / CT_VPR;
constexpr int CT_NPW = (CT_NPIX + CT_PSTEP - 1) / CT_PSTEP, CT_NPY = CT_TH * TW / CT_PSTEP;
constexpr int CT_XB = CT_NPW * CT_PSTEP * CT_PITCH, CT_YB = CT_NPY * CT_PSTEP * CT_PITCH;
constexpr int CT_LDS = 2 * (CT_XB + CT_YB);
static_assert(CT_NPW == 6 && CT_NPY == 4, "filler schedule below");

// MFMA slots of a plane before window row wr (rows carry 2, 4 x 7, 2 MFMAs)
__host__ __device__ constexpr int ct_slots_before(int wr) { return wr == 0 ? 0 : 2 + 4 * (wr - 1); }
}  // namespace

template <typename T>
__global__ __launch_bounds__(NTHR) void wgrad_convt16_kernel(const WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int PITCH = CT_PITCH, VPR = CT_VPR, PSTEP = CT_PSTEP, NPW = CT_NPW, NPY = CT_NPY, RW = CT_RW;
  constexpr int VEC = Tr<T>::VEC;
  typedef WindowStager<T, 128, NTHR, PITCH> Stager;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wq_m = wave >> 1, wq_n = wave & 1;
  // XCD placement of wgrad_mfma.hip: the co-tile workgroups of one (ci tile, pixel slice) share the window, consecutive groups go to one XCD
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  {
    const int gx_ = gridDim.x, gy_ = gridDim.y, gz_ = gridDim.z;
    const int groups = gx_ * gz_;
    const int lid = bx + gx_ * (by + gy_ * bz);
    if ((groups & 7) == 0) {
      const int x = lid & 7, q = lid >> 3;
      const int k = q / gy_, r = q - k * gy_;
      const int g = x * (groups >> 3) + k;
      by = r; bz = g / gx_; bx = g - bz * gx_;
    }
  }
  const int ci0 = bx * 64, co0 = by * 64;
  const int tiles_x = (a.OW + TW - 1) / TW, tiles_y = (a.OH + CT_TH - 1) / CT_TH;
  const int ntiles = a.N * tiles_x * tiles_y;

  char* const ldsX0 = smem;
  char* const ldsY0 = smem + 2 * CT_XB;

  f32x16_t acc[16];
#pragma unroll
  for (int t = 0; t < 16; ++t)
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[t][k] = 0.f;

  // per-lane operand addressing (wgrad_mfma.hip, 2-byte form)
  int ya0, xa0;
  {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3, hh = g >> 1, cb = g & 1;
    const int tx = 8 * hh + q;
    ya0 = tx * PITCH + (wq_m * 32 + 16 * cb + 4 * p) * 2;
    xa0 = tx * PITCH + (wq_n * 32 + 16 * cb + 4 * p) * 2;
  }

  Stager sg;
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
    tp.y0 = tyi * CT_TH; tp.x0 = (rem - tyi * tiles_x) * TW;
    return tp;
  };
  const int dy_pix_bytes = a.dyC * (int)sizeof(T), dy_row_bytes = a.DW * dy_pix_bytes;
  auto y_base = [&](int n) { return (const char*)a.dy + ((size_t)n * a.DH * a.DW * a.dyC + (ycok ? yc : 0)) * sizeof(T); };
  // dy pixel of low-resolution grid position (gy, gx) in parity plane (py, px): (2 gy + py, 2 gx + px); clamped address, validity at the store
  auto load_y = [&](const char* ybase, const TilePos& tp, int ty, int tx, int py, int px, bool& ok) {
    const int gy = tp.y0 + ty, gx = tp.x0 + tx;
    ok = ycok && gy < a.OH && gx < a.OW;
    const int gyc = min(gy, a.OH - 1), gxc = min(gx, a.OW - 1);
    const unsigned off = (unsigned)((gyc * 2 + py) * dy_row_bytes + (gxc * 2 + px) * dy_pix_bytes);
    return *(const uint4*)(ybase + off);
  };

  uint4 yv[NPY], xv[NPW];
  bool yok[NPY], xok[NPW];
  int ypos[NPY], wpos[NPW];   // packed (row << 16 | col) of every pass, tile invariant
  bool win[NPW];
#pragma unroll
  for (int u = 0; u < NPY; ++u) {
    const int p = u * PSTEP + yp0;
    ypos[u] = ((p >> 4) << 16) | (p & 15);
  }
#pragma unroll
  for (int u = 0; u < NPW; ++u) {
    const int hp = u * PSTEP + sg.p0;
    const int hy = hp / RW;
    wpos[u] = (hy << 16) | (hp - hy * RW);
    win[u] = hp < CT_NPIX;
  }
  auto store_y = [&](char* oy, int u) { *(uint4*)(oy + (u * PSTEP + yp0) * PITCH + ycv * 16) = yok[u] ? yv[u] : make_uint4(0, 0, 0, 0); };

  int tile = bz;
  if (tile < ntiles) {
    // ---- prologue: window of tile 0 and plane (0, 0) into buffer 0; window of tile 1 and plane (0, 1) into the registers
    TilePos tp0 = tile_pos(tile);
    {
      const char* yb = y_base(tp0.n);
      sg.bind_image(tp0.n);
#pragma unroll
      for (int u = 0; u < NPY; ++u) yv[u] = load_y(yb, tp0, ypos[u] >> 16, ypos[u] & 0xffff, 0, 0, yok[u]);
#pragma unroll
      for (int u = 0; u < NPW; ++u) xv[u] = sg.load_at(wpos[u] >> 16, wpos[u] & 0xffff, win[u], tp0.y0 - 1, tp0.x0 - 1, 1, a.IH, a.IW, xok[u]);
#pragma unroll
      for (int u = 0; u < NPY; ++u) store_y(ldsY0, u);
#pragma unroll
      for (int u = 0; u < NPW; ++u) sg.write(ldsX0, u, xv[u], xok[u]);
      const int t1 = tile + a.ksplit < ntiles ? tile + a.ksplit : tile;
      const TilePos tp1 = tile_pos(t1);
      sg.bind_image(tp1.n);
#pragma unroll
      for (int u = 0; u < NPY; ++u) yv[u] = load_y(yb, tp0, ypos[u] >> 16, ypos[u] & 0xffff, 0, 1, yok[u]);
#pragma unroll
      for (int u = 0; u < NPW; ++u) xv[u] = sg.load_at(wpos[u] >> 16, wpos[u] & 0xffff, win[u], tp1.y0 - 1, tp1.x0 - 1, 1, a.IH, a.IW, xok[u]);
    }
    __syncthreads();

    struct Frag { s16x4_t lo, hi; };
    Frag xs[2][2], ys[4];
    // targets of this tile's fillers (set per tile): the tile whose planes 0 / 1 are loaded in planes 2 / 3, and the tile whose window is loaded
    TilePos tpn{0, 0, 0}, tpw{0, 0, 0};
    const char* ybn = nullptr;
    const char* ybc = nullptr;
    TilePos tpc{0, 0, 0};

    // one plane: P = 2 py + px.  bx_ / by_: window and dy buffers being contracted, oy / ox: the buffers being filled
    auto plane = [&](auto pc, const char* bxw, const char* byp, char* oy, char* ox) __attribute__((always_inline)) {
      constexpr int P = decltype(pc)::value, PY = P >> 1, PX = P & 1;
      const char* xbase = bxw + xa0;
      const char* ybase = byp + ya0;
      auto rd_x = [&](auto wc, auto sc, auto hc) __attribute__((always_inline)) {
        constexpr int Wn = decltype(wc)::value, S_ = decltype(sc)::value, H_ = decltype(hc)::value;
        constexpr int OFF = ((Wn + PY) * RW + S_ + PX) * PITCH + H_ * 4 * PITCH;
        if constexpr (H_ == 0) xs[Wn & 1][S_].lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(xbase + OFF));
        else xs[Wn & 1][S_].hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(xbase + OFF));
      };
      auto rd_y = [&](auto kc, auto hc) __attribute__((always_inline)) {
        constexpr int Kn = decltype(kc)::value, H_ = decltype(hc)::value;
        constexpr int OFF = Kn * TW * PITCH + H_ * 4 * PITCH;
        if constexpr (H_ == 0) ys[Kn & 3].lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(ybase + OFF));
        else ys[Kn & 3].hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(ybase + OFF));
      };
      auto prefetch = [&](auto wc, auto qc) __attribute__((always_inline)) {
        constexpr int Q = decltype(qc)::value;
        if constexpr (Q < 4) rd_x(wc, std::integral_constant<int, Q / 2>{}, std::integral_constant<int, Q % 2>{});
        else rd_y(wc, std::integral_constant<int, Q - 4>{});
      };
      // filler of MFMA slot S (0 .. 31) of this plane
      auto filler = [&](auto sc_) __attribute__((always_inline)) {
        constexpr int S = decltype(sc_)::value;
        // dy: stores of the next plane at slots 1, 4, 7, 10; loads of the plane after next at slots 17, 20, 23, 26
        if constexpr (S < 16 && S % 3 == 1 && S / 3 < NPY) store_y(oy, S / 3);
        if constexpr (S >= 17 && (S - 17) % 3 == 0 && (S - 17) / 3 < NPY) {
          constexpr int U = (S - 17) / 3;
          // plane after next: (this tile, P + 2) for P < 2, else (next tile, P - 2)
          if constexpr (P < 2) yv[U] = load_y(ybc, tpc, ypos[U] >> 16, ypos[U] & 0xffff, (P + 2) >> 1, (P + 2) & 1, yok[U]);
          else yv[U] = load_y(ybn, tpn, ypos[U] >> 16, ypos[U] & 0xffff, (P - 2) >> 1, (P - 2) & 1, yok[U]);
        }
        // window: three items per plane at slots 3, 13, 24 -- planes 0, 1 store tile k + 1 (held in registers), planes 2, 3 load tile k + 2
        if constexpr (S == 3 || S == 13 || S == 24) {
          constexpr int I = (P & 1) * 3 + (S == 3 ? 0 : S == 13 ? 1 : 2);
          if constexpr (P < 2) sg.write(ox, I, xv[I], xok[I]);
          else xv[I] = sg.load_at(wpos[I] >> 16, wpos[I] & 0xffff, win[I], tpw.y0 - 1, tpw.x0 - 1, 1, a.IH, a.IW, xok[I]);
        }
      };
      auto wrow = [&](auto wc) __attribute__((always_inline)) {
        constexpr int Wr = decltype(wc)::value;                     // window row of this plane, 0 .. 8
        constexpr int KLO = Wr - 1 < 0 ? 0 : Wr - 1, KHI = Wr < CT_TH ? Wr : CT_TH - 1;
        constexpr int CNT = (KHI - KLO + 1) * 2;
        constexpr int S0 = ct_slots_before(Wr);
        constexpr int NR = Wr + 1 > CT_TH ? 0 : (Wr + 1 < CT_TH ? 6 : 4);   // reads for window row Wr + 1 (dy rows end at 7)
        auto one = [&](auto jc) __attribute__((always_inline)) {
          constexpr int j = decltype(jc)::value;
          if constexpr (j < CNT) {
            constexpr int kk = KLO + j / 2, sft = j % 2, t = (1 - (Wr - kk)) * 2 + (1 - sft);
            Tr<T>::mma(__builtin_bit_cast(uint4, ys[kk & 3]), __builtin_bit_cast(uint4, xs[Wr & 1][sft]), acc[P * 4 + t]);
            __builtin_amdgcn_sched_barrier(0);
            constexpr int R0 = j * NR / CNT, R1 = (j + 1) * NR / CNT;
            if constexpr (R0 < R1) prefetch(std::integral_constant<int, Wr + 1>{}, std::integral_constant<int, R0>{});
            if constexpr (R0 + 1 < R1) prefetch(std::integral_constant<int, Wr + 1>{}, std::integral_constant<int, R0 + 1>{});
            if constexpr (R0 + 2 < R1) prefetch(std::integral_constant<int, Wr + 1>{}, std::integral_constant<int, R0 + 2>{});
            filler(std::integral_constant<int, S0 + j>{});
            __builtin_amdgcn_sched_barrier(0);
          }
        };
        one(std::integral_constant<int, 0>{}); one(std::integral_constant<int, 1>{});
        one(std::integral_constant<int, 2>{}); one(std::integral_constant<int, 3>{});
      };
      {
        auto z = std::integral_constant<int, 0>{};
        auto o = std::integral_constant<int, 1>{};
        rd_y(z, z); rd_y(z, o);
        rd_x(z, z, z); rd_x(z, z, o);
        rd_x(z, o, z); rd_x(z, o, o);
      }
      wrow(std::integral_constant<int, 0>{}); wrow(std::integral_constant<int, 1>{}); wrow(std::integral_constant<int, 2>{});
      wrow(std::integral_constant<int, 3>{}); wrow(std::integral_constant<int, 4>{}); wrow(std::integral_constant<int, 5>{});
      wrow(std::integral_constant<int, 6>{}); wrow(std::integral_constant<int, 7>{}); wrow(std::integral_constant<int, 8>{});
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
    };

    int cur = 0;
    for (; tile < ntiles; tile += a.ksplit) {
      const int t1 = tile + a.ksplit < ntiles ? tile + a.ksplit : tile;          // clamped: a re-fetch at the tail is harmless
      const int t2 = tile + 2 * a.ksplit < ntiles ? tile + 2 * a.ksplit : tile;
      tpc = tile_pos(tile); ybc = y_base(tpc.n);
      tpn = tile_pos(t1); ybn = y_base(tpn.n);
      tpw = tile_pos(t2);
      const char* bxw = ldsX0 + cur * CT_XB;
      char* ox = ldsX0 + (cur ^ 1) * CT_XB;
      char* y0b = ldsY0;
      char* y1b = ldsY0 + CT_YB;
      // (four planes per tile: plane 0 is always in dy buffer 0)
      plane(std::integral_constant<int, 0>{}, bxw, y0b, y1b, ox);
      plane(std::integral_constant<int, 1>{}, bxw, y1b, y0b, ox);
      sg.bind_image(tpw.n);   // the window loads of planes 2, 3 address tile k + 2 (the stores of planes 0, 1 only needed scale / shift)
      plane(std::integral_constant<int, 2>{}, bxw, y0b, y1b, ox);
      plane(std::integral_constant<int, 3>{}, bxw, y1b, y0b, ox);
      cur ^= 1;
    }
  }

  // ---- combine: fp32 atomics, lanes 0-31 cover 128 contiguous bytes of one dW row.  Accumulator P * 4 + t holds window offset
  // (wy, wx) = (1 - t / 2, 1 - t % 2) of parity (py, px) = (P / 2, P % 2): kernel tap r = 3 - py - 2 wy, s = 3 - px - 2 wx
  const int ci = ci0 + wq_n * 32 + (lane & 31);
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int P = q >> 2, t = q & 3;
    const int r = 3 - (P >> 1) - 2 * (1 - (t >> 1)), s = 3 - (P & 1) - 2 * (1 - (t & 1));
    const int tw = r * 4 + s;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int co = co0 + wq_m * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
      if (co < a.Cout && ci < a.Cin) atomicAdd(a.dW + ((size_t)tw * a.Cout + co) * a.Cin + ci, acc[q][i]);
    }
  }
}

// `a`: any of the four parity launches of wgrad_launches' transposed form (sources, extents, dy, dW are the same in all four)
bool wgrad_convt16_eligible(const WgradArgs& a, int dtype) {
  static const bool off = getenv("OCTSEG_NO_WGRAD_CONVT16") != nullptr;   // A/B switch: four per-parity launches
  if (off || dtype != DT_BF16) return false;
  if (a.ntaps != 4 || a.istride != 1 || a.dstride != 2 || a.span_x != 2 || a.span_y != 2) return false;
  if (a.OH != a.IH || a.OW != a.IW || a.DH != 2 * a.IH || a.DW != 2 * a.IW) return false;
  if (a.Cin % 8 != 0 || a.dyC % 8 != 0 || a.Cin < 32 || a.Cout < 32) return false;
  for (int i = 0; i < a.nsrc; ++i)
    if (a.src[i].up || a.src[i].H != a.IH || a.src[i].W != a.IW) return false;
  // (the staged image offsets are 32-bit)
  if ((long long)a.DH * a.DW * a.dyC * 2 >= (1ll << 31)) return false;
  return (long long)a.N * ((a.OH + CT_TH - 1) / CT_TH) * ((a.OW + TW - 1) / TW) >= 2;
}

hipError_t launch_wgrad_convt16(int dtype, const WgradArgs& a0, hipStream_t st) {
  if (!wgrad_convt16_eligible(a0, dtype)) return hipErrorInvalidValue;
  WgradArgs a = a0;
  const int ntiles = a.N * ((a.OW + TW - 1) / TW) * ((a.OH + CT_TH - 1) / CT_TH);
  const int gx = (a.Cin + 63) / 64, gy = (a.Cout + 63) / 64;
  static const int wg_env = getenv("OCTSEG_WGRAD_WGS") ? atoi(getenv("OCTSEG_WGRAD_WGS")) : 0;   // experiments
  int ks = (wg_env > 0 ? wg_env : (a.wg_target > 0 ? a.wg_target : 256)) / (gx * gy);
  if (ks > ntiles) ks = ntiles;
  if (ks < 1) ks = 1;
  if (deterministic_mode()) ks = 1;   // one writer per dW element: a fixed summation order
  a.ksplit = ks;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)wgrad_convt16_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL((wgrad_convt16_kernel<bf16_t>), dim3(gx, gy, ks), dim3(NTHR), CT_LDS, st, a);
  return hipGetLastError();
}

}  // namespace octseg
