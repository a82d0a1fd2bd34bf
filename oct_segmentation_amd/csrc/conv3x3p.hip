// conv3x3p.hip -- persistent 3x3 stride-1 convolution (forward and data gradient) for the 2-byte dtypes on gfx950.
//
// Why: conv_mfma_kernel runs ONE output tile per workgroup and one workgroup per CU (141 KiB of LDS).  Its main loop holds
// ~1.2 PFLOP/s (slope of time over K), but every tile round also pays ~26 us that no MFMA covers: all 256 CUs fill their first
// window + slabs from HBM at the same moment, and drain their 64 KiB output tile at the same moment (s_memtime stamps, 512 ->
// 256 @176^2: prologue 23.5 k + epilogue 18.4 k ticks against 210 k of main loop; on the K-thin data gradients the two exceed
// the main loop).  Here a workgroup is persistent: it walks its tiles with ONE software pipeline over the flattened
// (tile, K chunk, tap) sequence -- the first window and slabs of the next tile are staged during the last taps of this one,
// and the epilogue's global stores drain under the next tile's MFMAs.  Tile, wave layout, window pitch, weight image and
// the lazy-BN staging are those of conv_mfma_kernel's <T, 2, 2, 4, 128> configuration (16x16 pixels x 128 channels, 8 waves,
// 64-channel K chunks, grouped A-row map); the pipeline is plain HIP (hipcc places every wait):
//   tap t:  LDS-store the weight slab of step g+1 and the window slice of pass t-1 (both loaded during tap t-1),
//           issue the loads of the slab of step g+2 and of window pass t (next chunk: possibly the next tile's first),
//           16 MFMAs per wave from the current window / slab slot, one barrier.
// The epilogue transposes the tile through the window buffer it has just finished with, half a tile (128 pixels) at a time.
// Eligible launches (launch_conv routes them here): 9 taps, stride 1, interior tiles only (OH, OW multiples of 16), the
// 128-channel N tile, K >= 128, >= 2 tiles per workgroup, no NCHW head.  Everything else stays on conv_mfma_kernel.
// RESULT (round 2): results bit-identical to conv_mfma_kernel's (same accumulation order); speed about EQUAL overall -- the
// overlap of the next tile's fill buys what the plain-HIP main loop loses to the hand-scheduled one (run9r: LDS-DMA slabs,
// counted waits).  The structure is the base for the next round (LDS-DMA slab ring + epilogue stores overlapped as well).
#include "common.h"
#include "conv_common.h"
#include "kernels.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace octseg {

namespace {
constexpr int P3_NT = 2, P3_WN = 2, P3_WM = 4, P3_RB = 128, P3_BN = 128, P3_TH = 16;
constexpr int P3_NTHREADS = 512, P3_PITCH = P3_RB + 16, P3_SLAB = P3_BN * P3_RB;
constexpr int P3_RW = 18, P3_NPIX = 18 * 18, P3_PSTEP = P3_NTHREADS / (P3_RB / 16), P3_NPASS = (P3_NPIX + P3_PSTEP - 1) / P3_PSTEP;
constexpr int P3_ABYTES = P3_NPASS * P3_PSTEP * P3_PITCH;
constexpr int P3_OPITCH = P3_BN * 2 + 16;
static_assert(P3_NPASS <= 8, "one window pass per tap");
// grouped A-row map (conv_mfma.hip RowMap<true>): the 16 lanes of one ds_read_b128 LDS cycle read 16 consecutive pixels of a row
__device__ __forceinline__ int p3_ty(int rr) { return (0xF00F0FF0u >> rr) & 1; }
__device__ __forceinline__ int p3_tx(int rr) {
  const unsigned grp = ((0xF00F0FF0u >> rr) & 1) ? 0xF00F0FF0u : ~0xF00F0FF0u;
  return __popc(grp & ((1u << rr) - 1u));
}
}  // namespace

// ILV = false: weight slab and window slice of the next step go to LDS at the head of the tap and the loads of the step after are issued
// right behind them (two slab slots, fragments fetched at the head of the tap).  ILV = true (bf16 default): three slab slots, loads two
// taps ahead of their LDS stores (two register sets), the first two fragment sets of the NEXT tap fetched behind this tap's last MFMAs and
// kept in flight across the barrier, and the tap cut into sixteen fenced slots of {MFMA, fragment read, a piece of the staging code}
// (branch-free staging, slab cursor resolved at compile time).  Both are bit-identical; on one box they run within 2 % of each other,
// because on real operands the loop sits on the 1400 W package limit and its time follows the energy of a tile, not the order of its
// instructions (profiles/r2_conv3x3p_probe.txt, profiles/r2_power_probe.txt).
// NOAFF (ILV only): no source carries a lazy BatchNorm or ReLU (the data gradients: dy is a plain tensor) -- the staged window words go to
// LDS as loaded instead of through the identity affine (unpack, v_pk_fma, v_cvt_pk, v_pk_max: 20 vector instructions per tap and thread).
// Bit-identical by construction (x * 1 + 0 rounds to x).
template <typename T, bool ILV, bool NOAFF = false>
__global__ __launch_bounds__(P3_NTHREADS, 1) void conv3x3p_kernel(const ConvArgs a, const int n_nt, const int tiles_x, const int tiles_y,
                                                                   const int nitems, const int gsz) {
  constexpr int NT = P3_NT, WN = P3_WN, WM = P3_WM, RB = P3_RB, BN = P3_BN, PITCH = P3_PITCH, SLAB = P3_SLAB;
  constexpr int KC = RB / 2, KSTEPS = RB / 32, VPR = RB / 16, NPASS = P3_NPASS, RW = P3_RW, OPITCH = P3_OPITCH;
  typedef WindowStager<T, RB, P3_NTHREADS> Stager;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ldsA = smem;                       // [2] windows
  char* ldsB = smem + 2 * P3_ABYTES;       // [2] weight slabs ([3] with ILV: slot = step % 3 = tap % 3)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;
  const int G = gridDim.x;
  const int wid = blockIdx.x;
  const int l = (G & 7) == 0 ? (wid & 7) * (G >> 3) + (wid >> 3) : wid;   // XCD x owns a contiguous range of the item order
  // Item order.  gsz == 0 (legacy): N tile fastest, M tile slowest, dealt to the workgroups round by round (item l + j G) -- every XCD
  // walks ALL N tiles all the time, and once their weights (9 K Cout 2 bytes) outgrow its 4 MB L2 every slab comes from beyond it
  // (768 <- 256 @176^2: 1.6 GB fetched per launch for 0.25 GB of operands, profiles/r2_traffic.json).  gsz > 0: the N tiles are cut
  // into groups of gsz whose weights fit an L2 next to the windows, the order is [group][M tile][N tile of the group], and XCD x owns
  // the x-th EIGHTH of that list outright (its 32 CUs walk it side by side): an XCD meets one or two groups in a whole launch.  The
  // price: the window of an M tile is fetched once per group instead of once.
  const bool grouped = gsz > 0 && (G & 7) == 0;
  const int cpx = G >> 3, seg = (nitems + 7) >> 3;
  const int seg0 = (wid & 7) * seg, seg1 = min(nitems, seg0 + seg), cu = wid >> 3;
  const int nloc = grouped ? max(0, (seg1 - seg0 - cu + cpx - 1) / cpx) : (nitems - l + G - 1) / G;
  auto item_of = [&](int j) -> int { return grouped ? seg0 + cu + j * cpx : l + j * G; };
  const int n_mt = nitems / n_nt, items_full = n_mt * (gsz > 0 ? gsz : 1), ngroups = gsz > 0 ? (n_nt + gsz - 1) / gsz : 1;
  auto split = [&](int it, int& m, int& nt) {
    if (!grouped) { nt = it % n_nt; m = it / n_nt; return; }
    const int g = min(it / items_full, ngroups - 1);
    const int itg = it - g * items_full, sz = min(gsz, n_nt - g * gsz);
    m = itg / sz;
    nt = g * gsz + itg - m * sz;
  };
  auto nt_of = [&](int it) -> int { int m, nt; split(it, m, nt); return nt; };
  const int nchunks = (a.Cin + KC - 1) / KC;
  const bool usrc = a.src_uniform != 0;

  struct Tile { int n, y0, x0, nt, w_mt; };
  auto tile_of = [&](int it) -> Tile {
    Tile t;
    int m;
    split(it, m, t.nt);
    t.w_mt = m;
    t.n = m / (tiles_x * tiles_y);
    m -= t.n * tiles_x * tiles_y;
    const int tyi = m / tiles_x;
    t.y0 = tyi * P3_TH; t.x0 = (m - tyi * tiles_x) * TW;
    return t;
  };

  // ---- fragment addressing
  int abase[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int ty = wm * 4 + mt * 2 + p3_ty(r), tx = p3_tx(r);
    abase[mt] = (ty * RW + tx) * PITCH + h * 16;
  }
  int bbase[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bbase[nt] = (wn * NT * 32 + nt * 32 + r) * RB;
  const int bswz = (((wn * NT * 32 + r) / 2) & (VPR - 1));   // (rows of one 32-channel sub-tile differ by multiples of 32: same swizzle)
  // tap tables in VGPR lanes (lane i = tap i), fetched with v_readlane
  int v_toff = 0, v_tapw = 0;
  if (lane < 9) {
    v_toff = ((a.tap_dy[lane] - a.min_dy) * RW + (a.tap_dx[lane] - a.min_dx)) * PITCH;
    v_tapw = a.tap_w[lane];
  }
  // window pixels of this thread's six passes (they do not depend on the tile)
  const int p0w = tid / VPR;
  // (window pixel of pass p: computed where it is used -- twelve registers of pass tables cost more than four VALU per tap)
  auto pass_y = [&](int p) __attribute__((always_inline)) {
    const int hp = min(p * P3_PSTEP + p0w, P3_NPIX - 1);   // (the padding rows of the last pass re-stage the last pixel: never read)
    return (hp * 3641) >> 16;                               // hp / 18 for hp < 324
  };
  auto pass_x = [&](int p) __attribute__((always_inline)) {
    const int hp = min(p * P3_PSTEP + p0w, P3_NPIX - 1);
    return hp - ((hp * 3641) >> 16) * RW;
  };

  f32x16_t acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;
  // ILV computes the same 64 x 64 wave tile with v_mfma_f32_16x16x32 (4 x 4 blocks of 16 x 16, 32 channels per MFMA): block row mb = tile row
  // wm * 4 + mb, lane & 15 = pixel of the row (A) / output channel of the block (B), lane >> 4 = 16-byte slice of the 64-byte k-step
  constexpr bool M16 = ILV;
  f32x4_t acc16[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int k = 0; k < 4; ++k) acc16[i][j][k] = 0.f;
  const int r16 = lane & 15, g16 = lane >> 4;
  int abase16[4], bbase16[4];
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) abase16[mb] = ((wm * 4 + mb) * RW + r16) * PITCH + g16 * 16;
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) bbase16[nb] = (wn * NT * 32 + nb * 16 + r16) * RB;
  const int bswz16 = (r16 >> 1) & (VPR - 1);   // (block rows differ by multiples of 16: the same swizzle)

  // ---- weight slab cursor: two steps ahead of the consumer
  const char* Wp = (const char*)a.W;
  const long long slab_stride = (long long)n_nt * SLAB;   // between consecutive (tap, chunk) slabs of one N tile
  int s_j = 0, s_chunk = 0, s_tap = 0, s_nt = nt_of(item_of(0));
  uint4 B0, B1;                                           // slab registers (ILV: the set of the even steps ...
  uint4 C0 = make_uint4(0, 0, 0, 0), C1 = C0;             // ... and of the odd ones: a slab stays in registers for two taps)
  auto load_slab_into = [&](uint4& b0, uint4& b1) __attribute__((always_inline)) {
    const int tapw = __builtin_amdgcn_readlane(v_tapw, s_tap);
    const char* p = Wp + ((long long)(tapw * nchunks + s_chunk)) * slab_stride + (long long)s_nt * SLAB + tid * 16;
    b0 = *(const uint4*)p; b1 = *(const uint4*)(p + 8192);
    if (++s_tap == 9) {
      s_tap = 0;
      if (++s_chunk == nchunks) {
        s_chunk = 0;
        if (s_j + 1 < nloc) { ++s_j; s_nt = nt_of(item_of(s_j)); }   // past the last item: stay (harmless re-fetch)
      }
    }
  };
  auto load_slab = [&]() __attribute__((always_inline)) { load_slab_into(B0, B1); };
  // ILV: the tap of the slab is known at compile time (consumer tap + 4), the wrap is branch-free
  const int g_mod = G % n_nt;
  auto load_slab_ct = [&](auto tic, uint4& b0, uint4& b1) __attribute__((always_inline)) {
    constexpr int TI = decltype(tic)::value;
    const int tapw = __builtin_amdgcn_readlane(v_tapw, TI);
    const char* p = Wp + ((long long)(tapw * nchunks + s_chunk)) * slab_stride + (long long)s_nt * SLAB + tid * 16;
    b0 = *(const uint4*)p; b1 = *(const uint4*)(p + 8192);
    if constexpr (TI == 8) {
      const bool wrap = s_chunk + 1 == nchunks;
      s_chunk = wrap ? 0 : s_chunk + 1;
      const bool more = wrap && s_j + 1 < nloc;
      s_j += more ? 1 : 0;
      if (grouped) {                       // (kernel-uniform branch)
        s_nt = nt_of(item_of(s_j));
      } else {
        const int nn = s_nt + g_mod;
        s_nt = more ? (nn >= n_nt ? nn - n_nt : nn) : s_nt;
      }
    }
  };
  auto store_slab_from = [&](int slot, const uint4& b0, const uint4& b1) __attribute__((always_inline)) {
    char* q = ldsB + slot * SLAB + tid * 16;
    *(uint4*)q = b0; *(uint4*)(q + 8192) = b1;
  };
  auto store_slab = [&](int slot) __attribute__((always_inline)) { store_slab_from(slot, B0, B1); };

  // ---- window producer: one chunk ahead of the consumer
  Stager nxt;
  int n_gy0 = 0, n_gx0 = 0;
  auto setup_next = [&](const Tile& t, int chunk) __attribute__((always_inline)) {
    nxt.setup(a.src, a.nsrc, a.Cin, chunk, tid, usrc);
    nxt.bind_image(t.n);
    n_gy0 = t.y0 + a.min_dy; n_gx0 = t.x0 + a.min_dx;
  };
  uint4 sl = make_uint4(0, 0, 0, 0), sl2 = sl;   // window passes in flight (ILV: even / odd taps, stored two taps after their load)
  bool sl_ok = false, sl2_ok = false;

  // ---- prologue: window of (first tile, chunk 0), slab of step 0 in slot 0, slab of step 1 in registers
  Tile cur = tile_of(item_of(0));
  {
    setup_next(cur, 0);
    load_slab();
    uint4 v[NPASS]; bool ok[NPASS];
#pragma unroll
    for (int p = 0; p < NPASS; ++p) v[p] = nxt.load_at(pass_y(p), pass_x(p), true, n_gy0, n_gx0, 1, a.IH, a.IW, ok[p]);
    store_slab(0);
    load_slab();                              // slab of step 1
    if constexpr (ILV) {                      // step 1 published too; steps 2 and 3 in the even / odd register set (stored during steps 0 / 1)
      store_slab(1);
      load_slab();
      load_slab_into(C0, C1);
    }
#pragma unroll
    for (int p = 0; p < NPASS; ++p) nxt.write_at(ldsA + (p * P3_PSTEP + p0w) * PITCH, v[p], ok[p]);
  }
  __syncthreads();

  // fragment sets (three in rotation; with ILV they live across taps: k-step ks of tap TT sits in set (TT + ks) % 3)
  uint4 af[3][2], bf[3][NT];
  auto frag_load = [&](int buf, const char* win, int toff, const char* bsl, int ks) __attribute__((always_inline)) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) af[buf][mt] = *(const uint4*)(win + abase[mt] + toff + ks * 32);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bf[buf][nt] = *(const uint4*)(bsl + bbase[nt] + (((ks * 2 + h) ^ bswz) * 16));
  };
  // M16: A fragments of the current k-step (one set, each replaced in place behind its last MFMA), B fragments of this and the next one
  uint4 fa[4], fb[2][4];
  auto ldA16 = [&](int mb, const char* win, int toff, int kq) __attribute__((always_inline)) {
    fa[mb] = *(const uint4*)(win + abase16[mb] + toff + kq * 64);
  };
  auto ldB16 = [&](int set, int nb, const char* bsl, int kq) __attribute__((always_inline)) {
    fb[set][nb] = *(const uint4*)(bsl + bbase16[nb] + (((kq * 4 + g16) ^ bswz16) * 16));
  };
  if constexpr (M16) {
    const int t0 = __builtin_amdgcn_readlane(v_toff, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) { ldA16(i, ldsA, t0, 0); ldB16(0, i, ldsB, 0); }
  } else if constexpr (ILV) {
    const int t0 = __builtin_amdgcn_readlane(v_toff, 0);
    frag_load(0, ldsA, t0, ldsB, 0);
    frag_load(1, ldsA, t0, ldsB, 1);
  }

  int jt = 0, chunk = 0, cb = 0;   // consumer: local item number, chunk, window buffer of the chunk
  Tile nt_tile = cur;              // tile of the chunk being staged

  // ---- epilogue of the finished tile `cur` through window buffer `buf` (every wave is past its last read of it)
  auto epilogue = [&](char* buf) __attribute__((always_inline)) {
    char* otile = buf;                                     // [128 pixels of a half tile][OPITCH]
    float* red = (float*)(buf + 128 * OPITCH);             // [WM][BN][2]
    const int co0 = cur.nt * BN;
    float s1[NT], s2[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { s1[nt] = 0.f; s2[nt] = 0.f; }
    float bias[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int co = co0 + wn * NT * 32 + nt * 32 + r;
      bias[nt] = (a.bias != nullptr && co < a.Cout) ? a.bias[co] : 0.f;
    }
    float bias16[4], t1[4], t2[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      const int co = co0 + wn * NT * 32 + nb * 16 + r16;
      bias16[nb] = (a.bias != nullptr && co < a.Cout) ? a.bias[co] : 0.f;
      t1[nb] = 0.f; t2[nb] = 0.f;
    }
    const float ofloor = a.relu_out ? 0.f : NO_FLOOR;
    // this thread's share of the store sweep: channel vector cvv, pixel (rloc, tx) of every wave-row group
    const int cvv = tid & 15, tx = (tid >> 4) & 15, rloc = tid >> 8;
    const int cov = co0 + cvv * 8;
    char* dptr = (char*)a.dst[0].ptr; int dC = a.dst[0].C, dc0 = a.dst[0].c0, dH = a.dst[0].H, dW = a.dst[0].W, dacc = a.dst[0].accum;
    int dpool = a.dst[0].pool;
#pragma unroll
    for (int i = 1; i < MAX_SRC; ++i)
      if (i < a.ndst && cov >= a.dst[i].c0) {
        dptr = (char*)a.dst[i].ptr; dC = a.dst[i].C; dc0 = a.dst[i].c0; dH = a.dst[i].H; dW = a.dst[i].W; dacc = a.dst[i].accum;
        dpool = a.dst[i].pool;
      }
    if (a.out_mode == OUT_ACCUM) dacc = 1;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      // transposed half tile into LDS
      if constexpr (M16) {
#pragma unroll
        for (int rl = 0; rl < 2; ++rl)       // row of the strip = block row 2 * mt + rl
#pragma unroll
          for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float val = clamp_lo(acc16[2 * mt + rl][nb][j] + bias16[nb], ofloor);
              t1[nb] += val; t2[nb] += val * val;
              acc16[2 * mt + rl][nb][j] = 0.f;
              const int q = wm * 32 + rl * 16 + 4 * g16 + j;
              *(unsigned short*)(otile + q * OPITCH + (wn * NT * 32 + nb * 16 + r16) * 2) = Tr<T>::bits16(val);
            }
      } else
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float val = clamp_lo(acc[mt][nt][i] + bias[nt], ofloor);
          s1[nt] += val; s2[nt] += val * val;
          acc[mt][nt][i] = 0.f;
          const int rl = h ^ ((0x6 >> (i >> 2)) & 1);     // grouped row map: row of the strip, column 4 * (i >> 2) + (i & 3)
          const int q = wm * 32 + rl * 16 + 4 * (i >> 2) + (i & 3);
          *(unsigned short*)(otile + q * OPITCH + (wn * NT * 32 + nt * 32 + r) * 2) = Tr<T>::bits16(val);
        }
      __syncthreads();
      if (cov < a.Cout) {
        if (dpool) {
          // gradient of a nearest-x2 upsample: each 2x2 quad summed in f32, one rounding, at half resolution
          if (rloc == 0 && (tx & 1) == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int ty = k * 4 + mt * 2;
              const char* l0 = otile + (k * 32 + tx) * OPITCH + cvv * 16;
              const uint4 qv[4] = {*(const uint4*)l0, *(const uint4*)(l0 + OPITCH), *(const uint4*)(l0 + 16 * OPITCH),
                                   *(const uint4*)(l0 + 17 * OPITCH)};
              uint4* gq = (uint4*)(dptr + ((((size_t)cur.n * dH + ((cur.y0 + ty) >> 1)) * dW + ((cur.x0 + tx) >> 1)) * dC + (cov - dc0)) * 2);
              uint4 old = make_uint4(0, 0, 0, 0);
              if (dacc) old = *gq;
              const unsigned o[4] = {old.x, old.y, old.z, old.w};
              const unsigned* qq[4] = {&qv[0].x, &qv[1].x, &qv[2].x, &qv[3].x};
              unsigned r4[4];
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                float lo = dacc ? Tr<T>::lo(o[i]) : 0.f, hi = dacc ? Tr<T>::hi(o[i]) : 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) { lo += Tr<T>::lo(qq[j][i]); hi += Tr<T>::hi(qq[j][i]); }
                r4[i] = Tr<T>::pk(lo, hi);
              }
              *gq = make_uint4(r4[0], r4[1], r4[2], r4[3]);
            }
          }
        } else {
          const size_t rowb = (size_t)dW * dC * 2;
          char* gp = dptr + ((((size_t)cur.n * dH + cur.y0 + mt * 2 + rloc) * dW + cur.x0 + tx) * dC + (cov - dc0)) * 2;
          const char* lp = otile + (rloc * 16 + tx) * OPITCH + cvv * 16;
          if (dacc) {
            uint4 old[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) old[k] = *(const uint4*)(gp + (size_t)k * 4 * rowb);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const uint4 val = *(const uint4*)(lp + k * 32 * OPITCH);
              float x0[8], x1[8];
              Tr<T>::unpack8(val, x0); Tr<T>::unpack8(old[k], x1);
#pragma unroll
              for (int e = 0; e < 8; ++e) x0[e] += x1[e];
              *(uint4*)(gp + (size_t)k * 4 * rowb) = Tr<T>::pack8(x0);
            }
          } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) *(uint4*)(gp + (size_t)k * 4 * rowb) = *(const uint4*)(lp + k * 32 * OPITCH);
          }
        }
      }
      __syncthreads();   // the half tile is consumed before the next half (or the next chunk's window slices) overwrites it
    }
    if (M16 && a.stat_slab != nullptr) {
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {   // the four lanes that share lane & 15 hold different pixels of one channel
        t1[nb] += __shfl_xor(t1[nb], 16); t1[nb] += __shfl_xor(t1[nb], 32);
        t2[nb] += __shfl_xor(t2[nb], 16); t2[nb] += __shfl_xor(t2[nb], 32);
        if (g16 == 0) {
          const int cl = wn * NT * 32 + nb * 16 + r16;
          red[(wm * BN + cl) * 2 + 0] = t1[nb];
          red[(wm * BN + cl) * 2 + 1] = t2[nb];
        }
      }
    }
    if (a.stat_slab != nullptr) {
      if constexpr (!M16)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        s1[nt] += __shfl_xor(s1[nt], 32);
        s2[nt] += __shfl_xor(s2[nt], 32);
        if (h == 0) {
          const int cl = wn * NT * 32 + nt * 32 + r;
          red[(wm * BN + cl) * 2 + 0] = s1[nt];
          red[(wm * BN + cl) * 2 + 1] = s2[nt];
        }
      }
      __syncthreads();
      if (tid < BN && co0 + tid < a.Cout) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) { t1 += red[(w * BN + tid) * 2]; t2 += red[(w * BN + tid) * 2 + 1]; }
        float* slab = a.stat_slab + ((size_t)(a.slab_row0 + cur.w_mt) * a.Cout + co0 + tid) * 2;
        slab[0] = t1; slab[1] = t2;
      }
      __syncthreads();
    }
  };

  // ---- one tap (ILV = false)
  auto tap = [&](auto tc, auto spc) __attribute__((always_inline)) {
    constexpr int TT = decltype(tc)::value;
    constexpr int SP = decltype(spc)::value;   // parity of the global step of tap 0 of this chunk (nine taps per chunk: it alternates)
    constexpr int P = (TT + SP) & 1;           // parity of this step = slab slot it reads
    char* awin = ldsA + cb * P3_ABYTES;
    char* anext = ldsA + (cb ^ 1) * P3_ABYTES;
    if constexpr (TT == 0) {   // which chunk is staged during this one: the next chunk of the tile, or the first of the next tile
      if (chunk + 1 < nchunks) setup_next(cur, chunk + 1);
      else { nt_tile = tile_of(item_of(jt + 1 < nloc ? jt + 1 : jt)); setup_next(nt_tile, 0); }
    }
    // what the previous tap loaded goes to LDS (the slab of the next step, window pass TT - 1 of the chunk being staged), then this
    // tap's loads are issued (the slab of step g + 2, window pass TT): a whole tap of MFMAs to land behind
    store_slab(P ^ 1);
    if constexpr (TT >= 1 && TT <= NPASS) nxt.write_at(anext + ((TT - 1) * P3_PSTEP + p0w) * PITCH, sl, sl_ok);
    load_slab();
    if constexpr (TT < NPASS) sl = nxt.load_at(pass_y(TT), pass_x(TT), true, n_gy0, n_gx0, 1, a.IH, a.IW, sl_ok);
    const int toff = __builtin_amdgcn_readlane(v_toff, TT);
    const char* bsl = ldsB + P * SLAB;
    frag_load(0, awin, toff, bsl, 0);
    frag_load(1, awin, toff, bsl, 1);
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      if (ks + 2 < KSTEPS) frag_load((ks + 2) % 3, awin, toff, bsl, ks + 2);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) Tr<T>::mma(af[ks % 3][mt], bf[ks % 3][nt], acc[mt][nt]);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  };

  // ---- one tap with v_mfma_f32_16x16x32 (M16): two k-steps of 16 MFMAs (block row major), 32 fenced slots.  Slot I = MFMA (mb, nb) of
  // k-step kq.  Behind it: slots 0, 1, 2, 4 of a k-step request the B fragments of the NEXT k-step (other set), slots 3, 7, 11, 15 the next
  // k-step's A fragment of the block row whose last MFMA was just issued (in place); the staging pieces sit in the other slots.
  auto tap16 = [&](auto tc, auto spc) __attribute__((always_inline)) {
    constexpr int TT = decltype(tc)::value;
    constexpr int SP = decltype(spc)::value;
    constexpr int P = (TT + SP) & 1;
    char* awin = ldsA + cb * P3_ABYTES;
    char* anext = ldsA + (cb ^ 1) * P3_ABYTES;
    if constexpr (TT == 0) {
      if (chunk + 1 < nchunks) setup_next(cur, chunk + 1);
      else { nt_tile = tile_of(item_of(jt + 1 < nloc ? jt + 1 : jt)); setup_next(nt_tile, 0); }
    }
    const int toff = __builtin_amdgcn_readlane(v_toff, TT);
    const char* bsl = ldsB + (TT % 3) * SLAB;
    const int toff_n = __builtin_amdgcn_readlane(v_toff, TT == 8 ? 0 : TT + 1);
    const char* awin_n = TT == 8 ? anext : awin;
    const char* bsl_n = ldsB + ((TT + 1) % 3) * SLAB;
    constexpr bool WST = TT >= 2 && TT <= NPASS + 1;
    constexpr bool WLD = TT < NPASS;
    uint4& wsl = (TT & 1) == 0 ? sl : sl2;
    bool& wok = (TT & 1) == 0 ? sl_ok : sl2_ok;
    uint4& sb0 = P == 0 ? B0 : C0;
    uint4& sb1 = P == 0 ? B1 : C1;
    unsigned wv[4] = {0u, 0u, 0u, 0u};
    const char* wptr = nullptr;
    bool wok_new = false;
    auto slot = [&](auto ic) __attribute__((always_inline)) {
      constexpr int I = decltype(ic)::value;
      constexpr int kq = I / 16, mb = (I % 16) / 4, nb = I % 4, J = I % 16;
      __builtin_amdgcn_sched_barrier(0);
      Tr<T>::mma16(fa[mb], fb[kq][nb], acc16[mb][nb]);
      {   // fragments of the next k-step: k-step 1 of this tap, or k-step 0 of the next one
        const char* win = kq == 0 ? awin : awin_n;
        const int tf = kq == 0 ? toff : toff_n;
        const char* bs = kq == 0 ? bsl : bsl_n;
        constexpr int kn = kq ^ 1;
        if constexpr (J == 0) ldB16(kn, 0, bs, kn);
        if constexpr (J == 1) ldB16(kn, 1, bs, kn);
        if constexpr (J == 2) ldB16(kn, 2, bs, kn);
        if constexpr (J == 4) ldB16(kn, 3, bs, kn);
        if constexpr (J % 4 == 3) ldA16(mb, win, tf, kn);
      }
      if constexpr (I == 5) *(uint4*)(ldsB + ((TT + 2) % 3) * SLAB + tid * 16) = sb0;
      if constexpr (I == 6) *(uint4*)(ldsB + ((TT + 2) % 3) * SLAB + tid * 16 + 8192) = sb1;
      if constexpr (WST && (I == 8 || I == 9 || I == 10 || I == 12)) {
        constexpr int k = I == 12 ? 3 : I - 8;
        const unsigned w = k == 0 ? wsl.x : k == 1 ? wsl.y : k == 2 ? wsl.z : wsl.w;
        if constexpr (NOAFF) wv[k] = w;
        else wv[k] = Tr<T>::affine_floor1(w, nxt.sc[2 * k], nxt.sc[2 * k + 1], nxt.sh[2 * k], nxt.sh[2 * k + 1], nxt.fl16);
      }
      if constexpr (WST && I == 13) {
        const uint4 v = wok ? make_uint4(wv[0], wv[1], wv[2], wv[3]) : make_uint4(0, 0, 0, 0);
        *(uint4*)(anext + ((TT - 2) * P3_PSTEP + p0w) * PITCH + nxt.cv * 16) = v;
      }
      if constexpr (I == 21) load_slab_ct(std::integral_constant<int, (TT + 4) % 9>{}, sb0, sb1);
      if constexpr (WLD && I == 24) wptr = nxt.addr_at(pass_y(TT), pass_x(TT), true, n_gy0, n_gx0, 1, a.IH, a.IW, wok_new);
      if constexpr (WLD && I == 25) { wsl = *(const uint4*)wptr; wok = wok_new; }
    };
#define P3_SLOT(i) slot(std::integral_constant<int, i>{})
    P3_SLOT(0); P3_SLOT(1); P3_SLOT(2); P3_SLOT(3); P3_SLOT(4); P3_SLOT(5); P3_SLOT(6); P3_SLOT(7);
    P3_SLOT(8); P3_SLOT(9); P3_SLOT(10); P3_SLOT(11); P3_SLOT(12); P3_SLOT(13); P3_SLOT(14); P3_SLOT(15);
    P3_SLOT(16); P3_SLOT(17); P3_SLOT(18); P3_SLOT(19); P3_SLOT(20); P3_SLOT(21); P3_SLOT(22); P3_SLOT(23);
    P3_SLOT(24); P3_SLOT(25); P3_SLOT(26); P3_SLOT(27); P3_SLOT(28); P3_SLOT(29); P3_SLOT(30); P3_SLOT(31);
#undef P3_SLOT
    __builtin_amdgcn_sched_barrier(0);
    // LDS operations retire in order: the LDS stores of slots 5, 6 and 13 are older than the eight fragment reads of the second k-step;
    // the two youngest of those (block rows 2 and 3 of the next tap, slots 27 and 31) stay in flight across the barrier
    asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- one tap, ILV: sixteen slots of {one MFMA, one fragment read, one piece of the staging code}, each fenced, so that the
  // staging instructions execute under the MFMA just issued instead of in blocks between the k-steps (see the kernel's header)
  auto tap_ilv = [&](auto tc, auto spc) __attribute__((always_inline)) {
    constexpr int TT = decltype(tc)::value;
    constexpr int SP = decltype(spc)::value;
    constexpr int P = (TT + SP) & 1;
    char* awin = ldsA + cb * P3_ABYTES;
    char* anext = ldsA + (cb ^ 1) * P3_ABYTES;
    if constexpr (TT == 0) {
      if (chunk + 1 < nchunks) setup_next(cur, chunk + 1);
      else { nt_tile = tile_of(item_of(jt + 1 < nloc ? jt + 1 : jt)); setup_next(nt_tile, 0); }
    }
    const int toff = __builtin_amdgcn_readlane(v_toff, TT);
    constexpr int R0 = TT % 3;
    const char* bsl = ldsB + (TT % 3) * SLAB;
    const int toff_n = __builtin_amdgcn_readlane(v_toff, TT == 8 ? 0 : TT + 1);
    const char* awin_n = TT == 8 ? anext : awin;
    const char* bsl_n = ldsB + ((TT + 1) % 3) * SLAB;
    constexpr bool WST = TT >= 2 && TT <= NPASS + 1;   // the window slice loaded two taps ago goes to LDS
    constexpr bool WLD = TT < NPASS;                   // window pass TT is loaded
    uint4& wsl = (TT & 1) == 0 ? sl : sl2;
    bool& wok = (TT & 1) == 0 ? sl_ok : sl2_ok;
    uint4& sb0 = P == 0 ? B0 : C0;
    uint4& sb1 = P == 0 ? B1 : C1;
    unsigned wv[4] = {0u, 0u, 0u, 0u};
    const char* wptr = nullptr;
    bool wok_new = false;
    auto slot = [&](auto ic) __attribute__((always_inline)) {
      constexpr int I = decltype(ic)::value;
      constexpr int ks = I / 4, mt = (I % 4) / 2, nt = I % 2, j = I % 4;
      __builtin_amdgcn_sched_barrier(0);
      Tr<T>::mma(af[(R0 + ks) % 3][mt], bf[(R0 + ks) % 3][nt], acc[mt][nt]);
      {   // one fragment of k-step ks + 2 (of the next tap past the end of this one)
        constexpr int k2 = (ks + 2) % KSTEPS, buf = (R0 + ks + 2) % 3;
        const char* win = ks + 2 < KSTEPS ? awin : awin_n;
        const int tf = ks + 2 < KSTEPS ? toff : toff_n;
        const char* bs = ks + 2 < KSTEPS ? bsl : bsl_n;
        if constexpr (j < 2) af[buf][j] = *(const uint4*)(win + abase[j] + tf + k2 * 32);
        else bf[buf][j - 2] = *(const uint4*)(bs + bbase[j - 2] + (((k2 * 2 + h) ^ bswz) * 16));
      }
      if constexpr (I == 0) *(uint4*)(ldsB + ((TT + 2) % 3) * SLAB + tid * 16) = sb0;
      if constexpr (I == 1) *(uint4*)(ldsB + ((TT + 2) % 3) * SLAB + tid * 16 + 8192) = sb1;
      if constexpr (WST && I >= 2 && I <= 5) {
        constexpr int k = I - 2;
        const unsigned w = k == 0 ? wsl.x : k == 1 ? wsl.y : k == 2 ? wsl.z : wsl.w;
        if constexpr (NOAFF) wv[k] = w;
        else wv[k] = Tr<T>::affine_floor1(w, nxt.sc[2 * k], nxt.sc[2 * k + 1], nxt.sh[2 * k], nxt.sh[2 * k + 1], nxt.fl16);
      }
      if constexpr (WST && I == 6) {
        const uint4 v = wok ? make_uint4(wv[0], wv[1], wv[2], wv[3]) : make_uint4(0, 0, 0, 0);
        *(uint4*)(anext + ((TT - 2) * P3_PSTEP + p0w) * PITCH + nxt.cv * 16) = v;
      }
      if constexpr (I == 7) load_slab_ct(std::integral_constant<int, (TT + 4) % 9>{}, sb0, sb1);
      if constexpr (WLD && I == 8) wptr = nxt.addr_at(pass_y(TT), pass_x(TT), true, n_gy0, n_gx0, 1, a.IH, a.IW, wok_new);
      if constexpr (WLD && I == 9) { wsl = *(const uint4*)wptr; wok = wok_new; }
    };
#define P3_SLOT(i) slot(std::integral_constant<int, i>{})
    P3_SLOT(0); P3_SLOT(1); P3_SLOT(2); P3_SLOT(3); P3_SLOT(4); P3_SLOT(5); P3_SLOT(6); P3_SLOT(7);
    P3_SLOT(8); P3_SLOT(9); P3_SLOT(10); P3_SLOT(11); P3_SLOT(12); P3_SLOT(13); P3_SLOT(14); P3_SLOT(15);
#undef P3_SLOT
    __builtin_amdgcn_sched_barrier(0);
    // LDS operations retire in order: everything but the four fragment reads of slots 12..15 has landed (this wave's LDS stores of
    // slots 0, 1 and 6 included), those four stay in flight across the barrier -- they read what an earlier barrier published
    asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  auto chunk_body = [&](auto spc) __attribute__((always_inline)) {
    auto one = [&](auto tc) __attribute__((always_inline)) {
      if constexpr (M16) tap16(tc, spc); else if constexpr (ILV) tap_ilv(tc, spc); else tap(tc, spc);
    };
    one(std::integral_constant<int, 0>{}); one(std::integral_constant<int, 1>{}); one(std::integral_constant<int, 2>{});
    one(std::integral_constant<int, 3>{}); one(std::integral_constant<int, 4>{}); one(std::integral_constant<int, 5>{});
    one(std::integral_constant<int, 6>{}); one(std::integral_constant<int, 7>{}); one(std::integral_constant<int, 8>{});
    if (chunk + 1 < nchunks) { ++chunk; }
    else {
      epilogue(ldsA + cb * P3_ABYTES);
      chunk = 0; ++jt; cur = nt_tile;
    }
    cb ^= 1;
  };
  const int total_chunks = nloc * nchunks;
  for (int cc = 0; cc < total_chunks; cc += 2) {
    chunk_body(std::integral_constant<int, 0>{});
    if (cc + 1 < total_chunks) chunk_body(std::integral_constant<int, 1>{});
  }
}

namespace {
struct P3Geom { int tiles_x, tiles_y, n_mt, n_nt, nitems, G, gsz; size_t lds; };
static P3Geom p3_geom(const ConvArgs& a) {
  P3Geom g;
  g.tiles_x = a.OW / TW; g.tiles_y = a.OH / P3_TH;
  g.n_mt = a.N * g.tiles_x * g.tiles_y;
  g.n_nt = (a.Cout + P3_BN - 1) / P3_BN;
  g.nitems = g.n_mt * g.n_nt;
  g.G = g.nitems < 256 ? g.nitems : 256;   // one workgroup per CU (gfx950 / MI355X only build)
  g.lds = (size_t)2 * P3_ABYTES + 2 * P3_SLAB;   // (+1 slab with ILV, added at launch)
  // N-tile groups (kernel header): only where the weights of all N tiles together outgrow an L2's share (2 MB); a group holds what fits
  // the budget.  OPT-IN (OCTSEG_P3_GROUP_KB=1536): measured on U-Net++/resnet101 16 x 704^2 in one run, the grouped order cuts the
  // kernel's TCC fetch from 1333 to 1146 MB per launch (-14 %, profiles/r3_p3_item_order.txt) and leaves its time where it was
  // (decoder data gradients 15.32 -> 15.43 ms per step, forward 13.70 -> 13.69: noise) -- the slabs it saves came from the 256 MB
  // Infinity Cache, not from HBM, and the loop's time follows the energy of its MFMAs.
  static const int group_kb = getenv("OCTSEG_P3_GROUP_KB") ? atoi(getenv("OCTSEG_P3_GROUP_KB")) : 0;   // A/B switch
  static const int all_kb = getenv("OCTSEG_P3_ALL_KB") ? atoi(getenv("OCTSEG_P3_ALL_KB")) : 2048;
  const long long per_nt = 9LL * a.Cin * P3_BN * 2;
  g.gsz = 0;
  if (group_kb > 0 && g.G == 256 && per_nt * g.n_nt > (long long)all_kb * 1024) {
    g.gsz = (int)std::max<long long>(1, (long long)group_kb * 1024 / per_nt);
    if (g.gsz >= g.n_nt) g.gsz = 0;
  }
  return g;
}
}  // namespace

bool conv3x3p_eligible(const ConvArgs& a, int dtype) {
  static const bool on = getenv("OCTSEG_NO_CONV3X3P") == nullptr;   // A/B switch
  if (!on || dtype == DT_F32) return false;
  if (a.ntaps != 9 || a.istride != 1 || a.ostride != 1 || a.ooy != 0 || a.oox != 0 || a.out_mode == OUT_HEAD_NCHW) return false;
  if (a.span_x != 3 || a.span_y != 3 || a.OH % P3_TH != 0 || a.OW % TW != 0 || a.IH != a.OH || a.IW != a.OW) return false;
  // the 128-channel N tile with 64-channel K chunks; K >= 128: measured on one box against conv_mfma_kernel (U-Net++/resnet101 16 x 704^2,
  // profiles/r2_conv3x3p_ab.txt) the wide data gradients gain (768 <- 256 @176^2 2000 -> 1813 us, 1024 <- 256 2586 -> 2525) and the
  // forwards 0..3 %, but the single-chunk (K = 64) data gradients lose 5..7 % to run9s, whose window never restages
  if (a.Cout <= 64 || a.Cin < 128 || a.Cout % 8 != 0) return false;
  for (int i = 0; i < a.ndst; ++i) {
    const DstDesc& d = a.dst[i];
    if (d.pool ? (d.H * 2 != a.OH || d.W * 2 != a.OW) : (d.H != a.OH || d.W != a.OW)) return false;
    if (d.c0 % 8 != 0) return false;
  }
  const ConvPackInfo pk = conv_pack_info(a, dtype);
  if (pk.BN != P3_BN || pk.RB != P3_RB) return false;                         // (the wide-N configuration keeps its layers)
  const long long items = (long long)a.N * (a.OH / P3_TH) * (a.OW / TW) * ((a.Cout + P3_BN - 1) / P3_BN);
  return items >= 512;                                                        // persistence needs at least two tiles per workgroup
}

int conv3x3p_rows(const ConvArgs& a) { return p3_geom(a).n_mt; }

template <typename T, bool ILV, bool NOAFF = false>
static hipError_t p3_launch_k(const ConvArgs& a, const P3Geom& g, hipStream_t st) {
  static bool set = false;
  if (!set) {
    hipError_t e = hipFuncSetAttribute((const void*)conv3x3p_kernel<T, ILV, NOAFF>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    set = true;
  }
  hipLaunchKernelGGL((conv3x3p_kernel<T, ILV, NOAFF>), dim3(g.G), dim3(P3_NTHREADS), g.lds + (ILV ? P3_SLAB : 0), st, a, g.n_nt, g.tiles_x, g.tiles_y,
                     g.nitems, g.gsz);
  return hipGetLastError();
}
template <typename T>
static hipError_t p3_launch(const ConvArgs& a, const P3Geom& g, hipStream_t st) {
  static const bool plain = getenv("OCTSEG_P3_PLAIN") != nullptr;   // A/B switch: the head-of-tap variant
  static const bool keep_aff = getenv("OCTSEG_P3_AFF") != nullptr;  // A/B switch: identity affine also where no source needs one
  if constexpr (std::is_same<T, f16_t>::value) return p3_launch_k<T, false>(a, g, st);   // (the interleaved variant's f16 instantiation spills)
  else {
    if (plain) return p3_launch_k<T, false>(a, g, st);
    bool noaff = !keep_aff;
    for (int i = 0; i < a.nsrc; ++i) noaff = noaff && a.src[i].scale == nullptr && a.src[i].shift == nullptr && a.src[i].relu == 0;
    return noaff ? p3_launch_k<T, true, true>(a, g, st) : p3_launch_k<T, true, false>(a, g, st);
  }
}

hipError_t launch_conv3x3p(int dtype, const ConvArgs& a0, hipStream_t st) {
  ConvArgs a = a0;
  a.src_uniform = 1;
  for (int i = 1; i < a.nsrc; ++i)
    if (a.src[i].c0 % 64 != 0) a.src_uniform = 0;
  const P3Geom g = p3_geom(a);
  if (dtype == DT_F16) return p3_launch<f16_t>(a, g, st);
  return p3_launch<bf16_t>(a, g, st);
}

}  // namespace octseg
