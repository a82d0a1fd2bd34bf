// conv_mfma.hip -- im2col-free NHWC convolution on gfx950 matrix cores.
//
// One kernel family serves every convolution of the hot path:
//   * forward 1x1 / 3x3 / 4x4 convs, stride 1 or 2 (tap table + istride),
//   * data gradients (same kernel, transposed weight pack, mirrored tap table),
//   * ConvTranspose2d 4x4 s2 (four parity launches, 2x2 taps each, ostride 2),
//   * the segmentation head (epilogue writes NCHW f32 logits + bias).
// The input is a *virtual* tensor: up to 5 concatenated sources, each optionally
// nearest-x2 upsampled and lazily batch-normalised (relu(x*scale+shift)) while it
// is staged into LDS -- torch.cat / F.interpolate / BN-apply / ReLU never touch HBM.
//
// Tiling: a workgroup of WM x WN waves owns a (4*WM) x 16 tile of output pixels x BN output
// channels; every wave computes 64 pixels x NT*32 channels from 32x32 MFMA tiles
// (v_mfma_f32_32x32x16_bf16, or 4x v_mfma_f32_32x32x2_f32 = exact f32 fmaf chain for the parity
// path).  Per channel chunk (RB bytes of K) the (tile + halo) input window is staged ONCE into LDS
// and every tap reads its A fragments from it at a shifted address (no im2col buffer anywhere);
// weights are pre-packed (pack_weight_image) into the exact, XOR-swizzled LDS image of every
// (tap, chunk, N-tile) slab and stream through a 2-slab LDS ring by LDS-DMA (global_load_lds, no
// VGPRs / ds_write / address math).  Software pipeline: while the MFMAs of (chunk c, tap t) run, the
// DMA of the next slab and the register-staged load of a slice of chunk c+1's
// window are in flight; one barrier per tap.  The epilogue adds bias, emits per-channel (sum, sumsq) partials for the
// following BatchNorm (deterministic slab, reduced by bn_finalize), transposes the accumulators
// through LDS and stores / accumulates whole 16-byte channel vectors.
#include "common.h"
#include "conv_common.h"
#include "kernels.h"

#include <type_traits>

namespace octseg {

#ifdef OCTSEG_STAMP
// diagnostic build: s_memtime stamps around the phases of the tap loop (never in the shipped library)
#define STAMP(var) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define STAMP(var) do { } while (0)
#endif

// A-row (0..31 of a 32x32 MFMA tile) -> pixel inside the wave's 2x16-pixel strip.  Plain: two rows of 16.
// Grouped: the 16 lanes that one ds_read_b128 LDS cycle serves ({0-3,12-15,20-27} / {4-11,16-19,28-31},
// MI355X_MICROARCH.md LDS table) read 16 CONSECUTIVE pixels of one row, which the pitch-128 XOR-swizzled
// window serves without bank conflicts for every tap shift.
template <bool GROUPED> struct RowMap;
template <> struct RowMap<false> {
  static __device__ __forceinline__ int ty(int rr) { return rr >> 4; }
  static __device__ __forceinline__ int tx(int rr) { return rr & 15; }
};
template <> struct RowMap<true> {
  static constexpr unsigned MASK_B = 0xF00F0FF0u;
  static __device__ __forceinline__ int ty(int rr) { return (MASK_B >> rr) & 1; }
  static __device__ __forceinline__ int tx(int rr) {
    const unsigned grp = ((MASK_B >> rr) & 1) ? MASK_B : ~MASK_B;
    return __popc(grp & ((1u << rr) - 1u));
  }
};

#ifdef OCTSEG_PLAIN_ROWMAP
constexpr bool ROWMAP_GROUPED = false;   // A/B build switch
#else
constexpr bool ROWMAP_GROUPED = true;
#endif

// M sub-tiles (2 rows x 16 pixels each) per wave: 2, or 4 in the wide-N configuration (NT = 4, 64-byte K chunks: four waves of
// 128 pixels x 128 channels, one per SIMD with the whole register file -- 256 accumulators)
static constexpr int wave_mt(int NT, int RB) { return (NT == 4 && RB == 64) ? 4 : 2; }

struct TilePos { int n, y0, x0, co0, w_mt, nt_idx; };

// XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (linear id % 8), each with a
// private 4 MiB L2; XCD x is given a contiguous range of the work order, so that what neighbours in that order
// share is read from HBM / MALL once per XCD.  Two orders (host choice, ConvArgs::tile_order):
//   N-major (0): an XCD streams ONE N tile's weights -- for weight sets that do not fit an L2 (the wide decoder layers);
//   M-major (1): the N tiles of one M tile are neighbours -- the activation window is fetched once instead of once
//                per N tile, and the (small) weight set is L2-resident on every XCD (ResNet bottleneck shapes).
// Speed only -- any placement is correct.
template <int TH_, int BN, bool T11 = false>
static __device__ __forceinline__ TilePos map_tile(const ConvArgs& a) {
  constexpr int TH = T11 ? 11 : TH_, TW = T11 ? 11 : octseg::TW;   // (T11: 11 x 11 pixel tiles, LOOP_T11)
  const int tiles_x = (a.OW + TW - 1) / TW, tiles_y = (a.OH + TH - 1) / TH;
  const int n_mt = gridDim.x, n_nt = gridDim.y;
  const int lid = blockIdx.x + blockIdx.y * n_mt;
  int mt_idx, nt_idx;
  {
    const int total = n_mt * n_nt;
    const int full = total & ~7;                              // the part of the grid that deals evenly over the XCDs
    int w = lid;
    if (lid < full) w = (lid & 7) * (full >> 3) + (lid >> 3);  // XCD x owns the work range [x, x+1) * full/8; the tail keeps the plain order
    if (a.tile_order) {   // M-major: the N tiles of one M tile are neighbours on one XCD -> its window comes from HBM once
      mt_idx = w / n_nt;
      nt_idx = w - mt_idx * n_nt;
    } else {              // N-major: an XCD streams one N tile's weights
      nt_idx = w / n_mt;
      mt_idx = w - nt_idx * n_mt;
    }
  }
  TilePos tp;
  tp.n = mt_idx / (tiles_x * tiles_y);
  mt_idx -= tp.n * tiles_x * tiles_y;
  const int tyi = mt_idx / tiles_x, txi = mt_idx - tyi * tiles_x;
  tp.y0 = tyi * TH; tp.x0 = txi * TW;
  tp.co0 = nt_idx * BN;
  tp.nt_idx = nt_idx;
  tp.w_mt = tp.n * tiles_x * tiles_y + mt_idx;   // M-tile index (BN-stat slab row)
  return tp;
}

// 2-byte types on the 64 x 64 wave tile issue v_mfma_f32_16x16x32 (the 128 x 128 wave tile of the wide-N configuration stays on 32x32x16: with
// 256 accumulators the re-mapped loop made hipcc spill the destinations of the asm loads, which the ISA audit rejects)
template <typename T> __host__ __device__ constexpr bool conv_m16(int mt) { return sizeof(T) == 2 && mt == 2; }

// Epilogue shared by the conv kernels: bias, BN partials, LDS transpose, 16-byte stores / accumulates.
// All waves must be past their last LDS read of the main loop (barrier) when this is entered.
template <typename T, int NT, int WN, int WM, bool GROUPED, int MT, bool T11 = false>
static __device__ __forceinline__ void conv_epilogue(const ConvArgs& a, char* smem, f32x16_t (&acc)[conv_m16<T>(MT) ? 1 : MT][conv_m16<T>(MT) ? 1 : NT],
                                                     f32x4_t (&acc16)[conv_m16<T>(MT) ? 2 * MT : 1][conv_m16<T>(MT) ? 2 * NT : 1], const TilePos& tp) {
  // 2-byte types accumulate in 16 x 16 blocks (v_mfma_f32_16x16x32: block row mb = tile row wm * 2 * MT + mb, element j of lane (r16, g) =
  // pixel 4 * g + j of that row, channel nb * 16 + r16); f32 in 32 x 32 blocks through the row map
  constexpr bool M16 = conv_m16<T>(MT);
  constexpr int NCB = M16 ? 2 * NT : NT;   // channel blocks of a wave, one channel of each per lane
  constexpr int NTHREADS = 64 * WM * WN;
  constexpr int BN = NT * 32 * WN;
  constexpr int TH = 2 * MT * WM;
  constexpr int BM = TH * TW;
  constexpr int VEC = Tr<T>::VEC;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;
  const int r16 = lane & 15, g16 = lane >> 4;
  const int n = tp.n, y0 = tp.y0, x0 = tp.x0, co0 = tp.co0, w_mt = tp.w_mt;
  // ---------------- epilogue ----------------
  // (all waves are past the last barrier: LDS is free)
  constexpr int OPITCH = BN * (int)sizeof(T) + 16;       // transposed-tile row pitch
  constexpr int OVPR = BN * (int)sizeof(T) / 16;         // 16-byte vectors per pixel row
  char* otile = smem;                                    // [BM][BN] T
  float* red = (float*)(smem + BM * OPITCH);             // [WM][BN][2] stat partials
  const bool head = a.out_mode == OUT_HEAD_NCHW;
  float s1[NCB], s2[NCB];
  {
    // Fast path (block-uniform): tile fully inside the output, one destination.  Everything below is the same
    // arithmetic in the same order as the general path, minus the per-element bounds checks, the destination
    // search and the per-vector 64-bit address math (the general path is ~4000 instructions per thread).
    if (!head && (T11 || (y0 + TH <= a.OH && x0 + TW <= a.OW))) {   // (11 x 11 tiles divide their maps: always whole)
      constexpr int ES = (int)sizeof(T);
      // accumulator element i of lane (r, h) is A row rr = (i & 3) + 8 * (i >> 2) + 4 * h; its pixel inside the wave's strip:
      //   plain row map:   row (i >> 3), column (i & 3) + 8 * ((i >> 2) & 1) + 4 * h
      //   grouped row map: row h ^ [(i >> 2) is 1 or 2], column 4 * (i >> 2) + (i & 3)      (RowMap<true>)
      const int lchan = (wn * NT * 32 + r) * ES;
      const int lbase = GROUPED ? (wm * 2 * MT * TW + h * TW) * OPITCH + lchan : (wm * 2 * MT * TW + 4 * h) * OPITCH + lchan;
      const int lbase_x = (wm * 2 * MT * TW + (1 - h) * TW) * OPITCH + lchan;   // grouped: the other row of the strip
      (void)lbase_x;
      // transposed tile into LDS; the bias add and the BN partial sums only where the layer has them (uniform
      // branches: a dgrad has neither and saves three of its four vector instructions per element)
      auto emit = [&](auto bias_c, auto stat_c, auto relu_c) __attribute__((always_inline)) {
        constexpr bool HAS_BIAS = decltype(bias_c)::value, HAS_STAT = decltype(stat_c)::value, HAS_RELU = decltype(relu_c)::value;
        if constexpr (M16) {
#pragma unroll
          for (int nb = 0; nb < NCB; ++nb) {
            s1[nb] = 0.f; s2[nb] = 0.f;
            const int cl = wn * NT * 32 + nb * 16 + r16;
            float bias = 0.f;
            if constexpr (HAS_BIAS) bias = co0 + cl < a.Cout ? a.bias[co0 + cl] : 0.f;
#pragma unroll
            for (int mb = 0; mb < 2 * MT; ++mb)
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                float val = acc16[mb][nb][j];
                if constexpr (HAS_BIAS) val += bias;
                if constexpr (HAS_RELU) val = clamp_lo(val, 0.f);
                if constexpr (HAS_STAT) {
                  if (!T11 || (wm * 2 * MT + mb) * TW + 4 * g16 + j < 121) { s1[nb] += val; s2[nb] += val * val; }   // (rows 121 .. 127 of an 11 x 11 tile repeat its last pixel)
                }
                *(unsigned short*)(otile + ((wm * 2 * MT + mb) * TW + 4 * g16 + j) * OPITCH + cl * ES) = Tr<T>::bits16(val);
              }
          }
        } else
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          s1[nt] = 0.f; s2[nt] = 0.f;
          float bias = 0.f;
          if constexpr (HAS_BIAS) {
            const int co = co0 + wn * NT * 32 + nt * 32 + r;
            bias = co < a.Cout ? a.bias[co] : 0.f;           // lanes past Cout: their sums are never stored
          }
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              float val = acc[mt][nt][i];
              if constexpr (HAS_BIAS) val += bias;
              if constexpr (HAS_RELU) val = clamp_lo(val, 0.f);   // eval, BatchNorm folded: the activation is stored, not the raw conv output
              if constexpr (HAS_STAT) { s1[nt] += val; s2[nt] += val * val; }
              int off, lb;
              if constexpr (GROUPED) {
                off = (mt * 2 * TW + 4 * (i >> 2) + (i & 3)) * OPITCH + nt * 32 * ES;
                lb = ((0x6 >> (i >> 2)) & 1) ? lbase_x : lbase;
              } else {
                off = ((mt * 2 + (i >> 3)) * TW + (i & 3) + 8 * ((i >> 2) & 1)) * OPITCH + nt * 32 * ES;
                lb = lbase;
              }
              if (sizeof(T) == 4) *(float*)(otile + lb + off) = val;
              else *(unsigned short*)(otile + lb + off) = Tr<T>::bits16(val);
            }
          }
        }
      };
      using std::true_type; using std::false_type;
      if (a.relu_out) emit(true_type{}, false_type{}, true_type{});   // (host: relu_out comes with the folded bias, never with statistics)
      else if (a.bias != nullptr) { if (a.stat_slab != nullptr) emit(true_type{}, true_type{}, false_type{}); else emit(true_type{}, false_type{}, false_type{}); }
      else { if (a.stat_slab != nullptr) emit(false_type{}, true_type{}, false_type{}); else emit(false_type{}, false_type{}, false_type{}); }
      if (a.stat_slab != nullptr) {
        if constexpr (M16) {
#pragma unroll
          for (int nb = 0; nb < NCB; ++nb) {   // the four lanes that share lane & 15 hold different pixels of one channel
            s1[nb] += __shfl_xor(s1[nb], 16); s1[nb] += __shfl_xor(s1[nb], 32);
            s2[nb] += __shfl_xor(s2[nb], 16); s2[nb] += __shfl_xor(s2[nb], 32);
            if (g16 == 0) {
              const int cl = wn * NT * 32 + nb * 16 + r16;
              red[(wm * BN + cl) * 2 + 0] = s1[nb];
              red[(wm * BN + cl) * 2 + 1] = s2[nb];
            }
          }
        } else
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
      }
      __syncthreads();
      if (a.stat_slab != nullptr && tid < BN && co0 + tid < a.Cout) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) { t1 += red[(w * BN + tid) * 2]; t2 += red[(w * BN + tid) * 2 + 1]; }
        float* slab = a.stat_slab + ((size_t)(a.slab_row0 + w_mt) * a.Cout + co0 + tid) * 2;
        slab[0] = t1; slab[1] = t2;
      }
      constexpr int PPI = NTHREADS / OVPR;                  // pixels stored per sweep of the workgroup
      static_assert(BM % PPI == 0 && (PPI % TW == 0 || TW % PPI == 0), "store sweep must tile the block");
      const int p = tid / OVPR, cvv = tid % OVPR;
      const int cov = co0 + cvv * VEC;                      // this thread's channel vector: the same in every sweep
      if (cov >= a.Cout) return;
      char* dptr = (char*)a.dst[0].ptr; int dC = a.dst[0].C, dc0 = a.dst[0].c0, dH = a.dst[0].H, dW = a.dst[0].W, dacc = a.dst[0].accum;
      int dpool = a.dst[0].pool;
#pragma unroll
      for (int i = 1; i < MAX_SRC; ++i)
        if (i < a.ndst && cov >= a.dst[i].c0) {
          dptr = (char*)a.dst[i].ptr; dC = a.dst[i].C; dc0 = a.dst[i].c0; dH = a.dst[i].H; dW = a.dst[i].W; dacc = a.dst[i].accum;
          dpool = a.dst[i].pool;
        }
      if (dpool) {
        // gradient of a nearest-x2 upsample: sum each 2x2 quad of the tile (T-rounded values, f32 sum, one rounding:
        // bit for bit what pool2x2_accum over a T temp gives) and store / accumulate at half resolution
        for (int k = 0; k < BM / PPI; ++k) {
          const int pp = p + k * PPI;
          const int ty = pp >> 4, tx = pp & 15;
          if ((ty | tx) & 1) continue;
          const char* l0 = otile + pp * OPITCH + cvv * 16;
          const uint4 q[4] = {*(const uint4*)l0, *(const uint4*)(l0 + OPITCH), *(const uint4*)(l0 + TW * OPITCH),
                              *(const uint4*)(l0 + (TW + 1) * OPITCH)};
          uint4* gq = (uint4*)(dptr + ((((size_t)n * dH + ((y0 + ty) >> 1)) * dW + ((x0 + tx) >> 1)) * dC + (cov - dc0)) * ES);
          uint4 old = make_uint4(0, 0, 0, 0);
          if (dacc) old = *gq;
          unsigned o[4] = {old.x, old.y, old.z, old.w}, r4[4];
          const unsigned* qq[4] = {&q[0].x, &q[1].x, &q[2].x, &q[3].x};
#pragma unroll
          for (int i = 0; i < 4; ++i) {   // old + q00 + q01 + q10 + q11 in f32, one rounding (the order of pool2x2_accum)
            if (sizeof(T) == 4) {
              float sacc = dacc ? __uint_as_float(o[i]) : 0.f;
#pragma unroll
              for (int j = 0; j < 4; ++j) sacc += __uint_as_float(qq[j][i]);
              r4[i] = __float_as_uint(sacc);
            } else {
              float lo = dacc ? Tr<T>::lo(o[i]) : 0.f, hi = dacc ? Tr<T>::hi(o[i]) : 0.f;
#pragma unroll
              for (int j = 0; j < 4; ++j) { lo += Tr<T>::lo(qq[j][i]); hi += Tr<T>::hi(qq[j][i]); }
              r4[i] = Tr<T>::pk(lo, hi);
            }
          }
          *gq = make_uint4(r4[0], r4[1], r4[2], r4[3]);
        }
        return;
      }
      if constexpr (T11) {   // rows of 11 pixels: the sweeps of 16 pixels do not follow the image rows -- one address per vector
        const char* lp = otile + p * OPITCH + cvv * 16;
        const bool accum = a.out_mode == OUT_ACCUM || dacc;
#pragma unroll
        for (int k = 0; k < BM / PPI; ++k) {
          const int pp = p + k * PPI;
          if (pp >= 121) break;
          const int ty = pp / 11, tx = pp - ty * 11;
          uint4* gq = (uint4*)(dptr + ((((size_t)n * dH + (y0 + ty)) * dW + (x0 + tx)) * dC + (cov - dc0)) * ES);
          uint4 val = *(const uint4*)(lp + k * PPI * OPITCH);
          if (accum) {
            const uint4 old = *gq;
            unsigned nv[4] = {val.x, val.y, val.z, val.w};
            const unsigned ov[4] = {old.x, old.y, old.z, old.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) nv[i] = Tr<T>::pk(Tr<T>::lo(nv[i]) + Tr<T>::lo(ov[i]), Tr<T>::hi(nv[i]) + Tr<T>::hi(ov[i]));
            val = make_uint4(nv[0], nv[1], nv[2], nv[3]);
          }
          *gq = val;
        }
        return;
      }
      const size_t pixb = (size_t)dC * ES * a.ostride;      // bytes between horizontally adjacent outputs
      const size_t rowb = (size_t)dW * dC * ES * a.ostride;
      const int oy = (y0 + (p >> 4)) * a.ostride + a.ooy, ox = (x0 + (p & 15)) * a.ostride + a.oox;
      char* gp = dptr + ((((size_t)n * dH + oy) * dW + ox) * dC + (cov - dc0)) * ES;
      // sweep k covers pixels p + k*PPI: whole rows further down, or (PPI < 16) the next piece of the same row
      auto gstep_of = [&](int k) -> size_t {
        if constexpr (PPI % TW == 0) return (size_t)(PPI / TW) * rowb;
        else return ((k + 1) % (TW / PPI) == 0) ? rowb - (size_t)(TW - PPI) * pixb : (size_t)PPI * pixb;
      };
      const char* lp = otile + p * OPITCH + cvv * 16;
      if (a.out_mode == OUT_ACCUM || dacc) {
#pragma unroll 4
        for (int k = 0; k < BM / PPI; ++k) {
          uint4 val = *(const uint4*)(lp + k * PPI * OPITCH);
          const uint4 old = *(const uint4*)gp;
          if (sizeof(T) == 4) {
            val.x = __float_as_uint(__uint_as_float(val.x) + __uint_as_float(old.x));
            val.y = __float_as_uint(__uint_as_float(val.y) + __uint_as_float(old.y));
            val.z = __float_as_uint(__uint_as_float(val.z) + __uint_as_float(old.z));
            val.w = __float_as_uint(__uint_as_float(val.w) + __uint_as_float(old.w));
          } else {
            unsigned nv[4] = {val.x, val.y, val.z, val.w};
            const unsigned ov[4] = {old.x, old.y, old.z, old.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float lo = Tr<T>::lo(nv[i]) + Tr<T>::lo(ov[i]);
              const float hi = Tr<T>::hi(nv[i]) + Tr<T>::hi(ov[i]);
              nv[i] = Tr<T>::pk(lo, hi);
            }
            val = make_uint4(nv[0], nv[1], nv[2], nv[3]);
          }
          *(uint4*)gp = val;
          gp += gstep_of(k);
        }
      } else {
#pragma unroll
        for (int k = 0; k < BM / PPI; ++k) {
          *(uint4*)gp = *(const uint4*)(lp + k * PPI * OPITCH);
          gp += gstep_of(k);
        }
      }
      return;
    }
  }
  if constexpr (M16) {
#pragma unroll
    for (int nb = 0; nb < NCB; ++nb) {
      s1[nb] = 0.f; s2[nb] = 0.f;
      const int cl = wn * NT * 32 + nb * 16 + r16;
      const int co = co0 + cl;
      const bool cok = co < a.Cout;
      const float bias = (a.bias != nullptr && cok) ? a.bias[co] : 0.f;
#pragma unroll
      for (int mb = 0; mb < 2 * MT; ++mb)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          int ty = wm * 2 * MT + mb, tx = 4 * g16 + j;
          const int prow = ty * TW + tx;          // A row of this element = its row in the transposed tile
          bool pv = true;
          if constexpr (T11) { pv = prow < 121; ty = pv ? prow / 11 : 0; tx = pv ? prow - ty * 11 : 0; }   // 11 x 11 pixels on 128 rows
          const int gy = y0 + ty, gx = x0 + tx;
          float val = acc16[mb][nb][j] + bias;
          if (a.relu_out) val = clamp_lo(val, 0.f);
          if (cok && pv && gy < a.OH && gx < a.OW) {
            s1[nb] += val; s2[nb] += val * val;
            if (head) {
              const DstDesc& d = a.dst[0];
              const int oy = gy * a.ostride + a.ooy, ox = gx * a.ostride + a.oox;
              ((float*)d.ptr)[(((size_t)n * a.Cout + co) * d.H + oy) * d.W + ox] = val;
            }
          }
          if (!head) *(unsigned short*)(otile + prow * OPITCH + cl * 2) = Tr<T>::bits16(val);
        }
    }
  } else
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    s1[nt] = 0.f; s2[nt] = 0.f;
    const int cl = wn * NT * 32 + nt * 32 + r;
    const int co = co0 + cl;
    const bool cok = co < a.Cout;
    const float bias = (a.bias != nullptr && cok) ? a.bias[co] : 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int rr = (i & 3) + 8 * (i >> 2) + 4 * h;   // A row of this accumulator element
        const int ty = wm * 2 * MT + mt * 2 + RowMap<GROUPED>::ty(rr), tx = RowMap<GROUPED>::tx(rr);
        const int gy = y0 + ty, gx = x0 + tx;
        float val = acc[mt][nt][i] + bias;
        if (a.relu_out) val = clamp_lo(val, 0.f);
        if (cok && gy < a.OH && gx < a.OW) {
          s1[nt] += val; s2[nt] += val * val;
          if (head) {
            const DstDesc& d = a.dst[0];
            const int oy = gy * a.ostride + a.ooy, ox = gx * a.ostride + a.oox;
            ((float*)d.ptr)[(((size_t)n * a.Cout + co) * d.H + oy) * d.W + ox] = val;
          }
        }
        if (!head) {
          if (sizeof(T) == 4) *(float*)(otile + (ty * TW + tx) * OPITCH + cl * 4) = val;
          else *(unsigned short*)(otile + (ty * TW + tx) * OPITCH + cl * 2) = Tr<T>::bits16(val);
        }
      }
    }
  }
  if (a.stat_slab != nullptr) {
    if constexpr (M16) {
#pragma unroll
      for (int nb = 0; nb < NCB; ++nb) {
        s1[nb] += __shfl_xor(s1[nb], 16); s1[nb] += __shfl_xor(s1[nb], 32);
        s2[nb] += __shfl_xor(s2[nb], 16); s2[nb] += __shfl_xor(s2[nb], 32);
        if (g16 == 0) {
          const int cl = wn * NT * 32 + nb * 16 + r16;
          red[(wm * BN + cl) * 2 + 0] = s1[nb];
          red[(wm * BN + cl) * 2 + 1] = s2[nb];
        }
      }
    } else
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
  }
  __syncthreads();
  if (a.stat_slab != nullptr && tid < BN) {
    const int co = co0 + tid;
    if (co < a.Cout) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) { t1 += red[(w * BN + tid) * 2]; t2 += red[(w * BN + tid) * 2 + 1]; }
      float* slab = a.stat_slab + ((size_t)(a.slab_row0 + (w_mt)) * a.Cout + co) * 2;
      slab[0] = t1; slab[1] = t2;
    }
  }
  if (!head) {
    // cooperative store: every thread moves whole 16-byte channel vectors of one pixel
    for (int v = tid; v < BM * OVPR; v += NTHREADS) {
      const int p = v / OVPR, cvv = v % OVPR;
      int ty = p >> 4, tx = p & 15;
      if constexpr (T11) { if (p >= 121) continue; ty = p / 11; tx = p - ty * 11; }
      const int gy = y0 + ty, gx = x0 + tx;
      const int co = co0 + cvv * VEC;
      if (gy >= a.OH || gx >= a.OW || co >= a.Cout) continue;
      char* dptr = (char*)a.dst[0].ptr; int dC = a.dst[0].C, dc0 = a.dst[0].c0, dH = a.dst[0].H, dW = a.dst[0].W, dacc = a.dst[0].accum;
      int dpool = a.dst[0].pool;
#pragma unroll
      for (int i = 1; i < MAX_SRC; ++i)
        if (i < a.ndst && co >= a.dst[i].c0) {
          dptr = (char*)a.dst[i].ptr; dC = a.dst[i].C; dc0 = a.dst[i].c0; dH = a.dst[i].H; dW = a.dst[i].W; dacc = a.dst[i].accum;
          dpool = a.dst[i].pool;
        }
      if (dpool) {   // 2x2 quad sum into the half-resolution destination (tile origins and extents are even: a quad is
                     // inside the image as a whole or not at all)
        if ((ty | tx) & 1) continue;
        const char* l0 = otile + p * OPITCH + cvv * 16;
        const uint4 q[4] = {*(const uint4*)l0, *(const uint4*)(l0 + OPITCH), *(const uint4*)(l0 + TW * OPITCH),
                            *(const uint4*)(l0 + (TW + 1) * OPITCH)};
        uint4* gq = (uint4*)(dptr + ((((size_t)n * dH + (gy >> 1)) * dW + (gx >> 1)) * dC + (co - dc0)) * sizeof(T));
        uint4 old = make_uint4(0, 0, 0, 0);
        if (dacc) old = *gq;
        unsigned o[4] = {old.x, old.y, old.z, old.w}, r4[4];
        const unsigned* qq[4] = {&q[0].x, &q[1].x, &q[2].x, &q[3].x};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (sizeof(T) == 4) {
            float sacc = dacc ? __uint_as_float(o[i]) : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) sacc += __uint_as_float(qq[j][i]);
            r4[i] = __float_as_uint(sacc);
          } else {
            float lo = dacc ? Tr<T>::lo(o[i]) : 0.f, hi = dacc ? Tr<T>::hi(o[i]) : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) { lo += Tr<T>::lo(qq[j][i]); hi += Tr<T>::hi(qq[j][i]); }
            r4[i] = Tr<T>::pk(lo, hi);
          }
        }
        *gq = make_uint4(r4[0], r4[1], r4[2], r4[3]);
        continue;
      }
      const int oy = gy * a.ostride + a.ooy, ox = gx * a.ostride + a.oox;
      uint4* gp = (uint4*)(dptr + ((((size_t)n * dH + oy) * dW + ox) * dC + (co - dc0)) * sizeof(T));
      uint4 val = *(const uint4*)(otile + p * OPITCH + cvv * 16);
      if (a.out_mode == OUT_ACCUM || dacc) {
        const uint4 old = *gp;
        if (sizeof(T) == 4) {
          val.x = __float_as_uint(__uint_as_float(val.x) + __uint_as_float(old.x));
          val.y = __float_as_uint(__uint_as_float(val.y) + __uint_as_float(old.y));
          val.z = __float_as_uint(__uint_as_float(val.z) + __uint_as_float(old.z));
          val.w = __float_as_uint(__uint_as_float(val.w) + __uint_as_float(old.w));
        } else {
          unsigned nv[4] = {val.x, val.y, val.z, val.w};
          const unsigned ov[4] = {old.x, old.y, old.z, old.w};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float lo = Tr<T>::lo(nv[i]) + Tr<T>::lo(ov[i]);
            const float hi = Tr<T>::hi(nv[i]) + Tr<T>::hi(ov[i]);
            nv[i] = Tr<T>::pk(lo, hi);
          }
          val = make_uint4(nv[0], nv[1], nv[2], nv[3]);
        }
      }
      *gp = val;
    }
  }
}

template <int RB> struct ConvCfg { static constexpr int PITCH = RB + 16, KSTEPS = RB / 32, VPR = RB / 16; };

// 4-wave variants must leave room for a second workgroup per CU (<= 256 registers): they serve the short layers
// whose prologue / epilogue only hides under another workgroup's MFMAs
// LOOP selects the main loop that is compiled into the instantiation (one kernel per loop: with all of them in one
// function hipcc gives up on the by-value argument struct and moves it to scratch, and every layer pays the register
// budget of the largest loop):  0 rolled tap loop (any tap table)   1 resident taps (one K chunk, small slabs)
//                               2 1x1 (one tap, four window passes per chunk)   3 run9r (3x3, slab ring)
//                               4 run9s (3x3, ONE K chunk: slab ring only -- the K-thin data gradients of the decoder)
//                               5 masked (per-source tap subsets, ConvArgs::taps_per_src: the rolled loop over each chunk's own taps, two window passes per tap)
//                               6 the rolled loop on 11 x 11 pixel tiles (121 of the 128 rows of the 8 x 16 variant's M tile; single window
//                                 buffer, two workgroups per CU): 3x3 layers on maps that are multiples of 11 but not of 16 -- every 88^2 / 44^2 /
//                                 22^2 map of a 704^2 frame -- whose 16-pixel tiling ends in a nearly empty round (44^2 at 16 frames: 288 workgroups
//                                 on 256 CUs; here 512 on 512 slots) or half-empty tiles (22^2: 47 % -> 94.5 %)
//                               7 masked, on 11 x 11 pixel tiles (the tied data gradient over the 44^2 / 22^2 low-resolution maps)
enum { LOOP_GENERIC = 0, LOOP_RESIDENT = 1, LOOP_1X1 = 2, LOOP_RUN9 = 3, LOOP_RUN9S = 4, LOOP_MASKED = 5, LOOP_T11 = 6, LOOP_MASKED_T11 = 7 };

template <typename T, int NT, int WN, int WM, int RB, int LOOP>
__global__ __launch_bounds__(64 * WM * WN, ((WM * WN == 4 && wave_mt(NT, RB) == 2) ? 2 : 1)) void conv_mfma_kernel(const ConvArgs a, const int mode) {
  constexpr int MT = wave_mt(NT, RB);
  const int dbuf = mode & 1;
  constexpr bool resident = LOOP == LOOP_RESIDENT;   // single K chunk + small slabs: every tap's weights stay in LDS, no per-tap DMA / barrier
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NTHREADS = 64 * WM * WN;
  constexpr int BN = NT * 32 * WN;
  constexpr int TH = 2 * MT * WM;
  constexpr int KC = RB / (int)sizeof(T);
  constexpr int PITCH = ConvCfg<RB>::PITCH, KSTEPS = ConvCfg<RB>::KSTEPS, VPR = ConvCfg<RB>::VPR;
  constexpr int NWAVES = WM * WN;
  constexpr int BBYTES = BN * RB;                             // one weight slab = its packed image
  constexpr int NDMA = BBYTES / 1024;                         // 1 KiB LDS-DMA pieces per slab
  constexpr int DPW = (NDMA + NWAVES - 1) / NWAVES;           // pieces issued per wave
  constexpr int SWZ_DIV = 256 / RB;                           // rows per 256-byte LDS bank line
  constexpr int MAXP = 4;                                     // window passes prefetched per tap
  typedef WindowStager<T, RB, NTHREADS> Stager;

#ifdef OCTSEG_STAMP
  unsigned long long k0 = 0, k1 = 0, k2 = 0, k3 = 0;
  STAMP(k0);
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;

  constexpr bool T11 = LOOP == LOOP_T11 || LOOP == LOOP_MASKED_T11;
  const TilePos tp = map_tile<TH, BN, T11>(a);
  const int n = tp.n, y0 = tp.y0, x0 = tp.x0, nt_idx = tp.nt_idx;

  // window geometry
  const bool single = a.ntaps == 1;
  const int lstride = single ? 1 : a.istride;   // LDS lookup stride
  const int smul = single ? a.istride : 1;      // staging coordinate multiplier
  constexpr int THe = T11 ? 11 : TH, TWe = T11 ? 11 : TW;
  const int RH = single ? THe : (THe - 1) * a.istride + a.span_y;
  const int RW = single ? TWe : (TWe - 1) * a.istride + a.span_x;
  const int npix = RH * RW;
  const int npass = (npix + Stager::PSTEP - 1) / Stager::PSTEP;
  const float inv_rw = 1.0f / (float)RW;
  const int gy0 = y0 * a.istride + a.min_dy, gx0 = x0 * a.istride + a.min_dx;

  const int abytes = npass * Stager::PSTEP * PITCH;   // rows padded to whole passes (unconditional stores)
  char* ldsA = smem;                                  // [1 or 2] windows
  char* ldsB = smem + (dbuf ? 2 : 1) * abytes;        // [2] weight slab ring ([3] in the run9 ring mode)

  constexpr bool M16 = conv_m16<T>(MT);
  f32x16_t acc[M16 ? 1 : MT][M16 ? 1 : NT];
#pragma unroll
  for (int i = 0; i < (M16 ? 1 : MT); ++i)
#pragma unroll
    for (int j = 0; j < (M16 ? 1 : NT); ++j)
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;
  // 2-byte types: the same wave tile as 2 MT x 2 NT blocks of v_mfma_f32_16x16x32 (64-byte k-steps): at equal cycles per FLOP the shape
  // holds a higher clock under the package power limit (MI355X_MICROARCH.md DVFS note 7; +4 % measured in conv3x3p)
  f32x4_t acc16[M16 ? 2 * MT : 1][M16 ? 2 * NT : 1];
#pragma unroll
  for (int i = 0; i < (M16 ? 2 * MT : 1); ++i)
#pragma unroll
    for (int j = 0; j < (M16 ? 2 * NT : 1); ++j)
#pragma unroll
      for (int k = 0; k < 4; ++k) acc16[i][j][k] = 0.f;
  const int r16 = lane & 15, g16 = lane >> 4;
  int abase16[2 * MT], bbase16[2 * NT], bswz16[2 * NT];
#pragma unroll
  for (int mb = 0; mb < 2 * MT; ++mb) {   // block row mb = tile row wm * 2 MT + mb: 16 lanes read 16 consecutive pixels, conflict-free at pitch RB + 16
    abase16[mb] = (((wm * 2 * MT + mb) * lstride) * RW + r16 * lstride) * PITCH + g16 * 16;
    if constexpr (T11) {   // A row p = 16 (wm 2 MT + mb) + r16 is pixel (p / 11, p % 11) of the 11 x 11 tile; rows 121 .. 127 re-read the last pixel
      const int p = min((wm * 2 * MT + mb) * 16 + r16, 120);
      const int pty = p / 11;
      abase16[mb] = (pty * RW + (p - pty * 11)) * PITCH + g16 * 16;
    }
  }
#pragma unroll
  for (int nb = 0; nb < 2 * NT; ++nb) {
    const int row = wn * NT * 32 + nb * 16 + r16;
    bbase16[nb] = row * RB;
    bswz16[nb] = (row / SWZ_DIV) & (VPR - 1);
  }

  int abase[MT], bbase[NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    // grouped row map: the 16 lanes one ds_read_b128 LDS cycle serves read 16 CONSECUTIVE pixels of one window row, whose
    // 144-byte pitch spreads them over all 16 sixteen-byte slots of the 256-byte LDS line for every tap shift and window
    // width (the plain map mixed lanes of both strip rows in one group: two 2-way conflicts per group at RW = 18, 30 % of
    // all LDS cycles of the 3x3 layers, profiles/r1_sq_counters_conv3x3_512_256_352.txt)
    const int ty = wm * 2 * MT + mt * 2 + RowMap<ROWMAP_GROUPED>::ty(r), tx = RowMap<ROWMAP_GROUPED>::tx(r);
    abase[mt] = ((ty * lstride) * RW + tx * lstride) * PITCH + h * 16;
  }
  int bswz[NT];  // XOR swizzle of the 16-byte chunk index inside a slab row (matches pack_weight_image)
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int row = wn * NT * 32 + nt * 32 + r;
    bbase[nt] = row * RB;
    bswz[nt] = (row / SWZ_DIV) & (VPR - 1);
  }

  const int nchunks = (a.Cin + KC - 1) / KC;
  const char* Wp = (const char*)a.W;

  const int ntiles_n = gridDim.y;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const unsigned ldsB_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ldsB;  // LDS byte address
  // LDS-DMA of a packed slab into ring slot `slot`.  Issued through inline asm: hipcc cannot prove that
  // a builtin LDS-DMA write does not alias the ds_reads that follow (runtime ring offsets) and would drain
  // vmcnt(0) right behind it; the asm form is invisible to its waitcnt pass and is retired by the explicit
  // s_waitcnt in front of the barrier instead.  Per-lane source = precomputed base + scalar slab offset.
  const char* dma_src0 = Wp + (size_t)nt_idx * BBYTES + (size_t)wave_u * DPW * 1024 + lane * 16;
  const size_t slab_stride = (size_t)ntiles_n * BBYTES;  // between consecutive (tap, chunk) slabs
  const bool dma_wave = NDMA % NWAVES == 0 || wave_u * DPW < NDMA;
  auto dmaB = [&](int slab_idx, int slot) {   // slab_idx = tapw * nchunks + chunk
    if (dma_wave) {
      const char* gsrc0 = dma_src0 + (size_t)slab_idx * slab_stride;
#pragma unroll
      for (int j = 0; j < DPW; ++j) {
        const char* gsrc = gsrc0 + j * 1024;
        const unsigned dst = __builtin_amdgcn_readfirstlane(ldsB_addr + slot * BBYTES + (wave_u * DPW + j) * 1024);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
      }
    }
  };
  // tap tables in VGPR lanes (lane i = tap i), fetched per iteration with v_readlane: a kernarg s_load in
  // the loop costs its full scalar-cache latency every iteration
  int v_toff = 0, v_tapw = 0;
  if (lane < a.ntaps) {
    v_toff = single ? 0 : ((a.tap_dy[lane] - a.min_dy) * RW + (a.tap_dx[lane] - a.min_dx)) * PITCH;
    v_tapw = a.tap_w[lane];
  }
  auto stage_full = [&](const Stager& sg, char* dst) {
    for (int p = 0; p < npass; p += MAXP) {
      uint4 v[MAXP];
      bool ok[MAXP];
#pragma unroll
      for (int u = 0; u < MAXP; ++u) v[u] = sg.load(min(p + u, npass - 1), n, gy0, gx0, smul, RW, npix, inv_rw, a.IH, a.IW, ok[u]);
#pragma unroll
      for (int u = 0; u < MAXP; ++u) sg.write(dst, min(p + u, npass - 1), v[u], ok[u]);
    }
  };
  // MFMAs of one tap.  The LDS fragment reads run two k-steps ahead of the MFMAs that consume them
  // (three register sets): LDS latency under load is several hundred cycles, a k-step of MFMAs is ~128.
  auto mma_tap = [&](const char* awin, const char* bsl, int toff) {
    if constexpr (M16) {
      // A fragments of the current 64-byte k-step (replaced in place behind the last MFMA of their block row), B fragments of both k-steps
      constexpr int KQ = RB / 64;
      uint4 fa[2 * MT], fb[KQ][2 * NT];
      auto ldA = [&](int mb, int kq) __attribute__((always_inline)) { fa[mb] = *(const uint4*)(awin + abase16[mb] + toff + kq * 64); };
      auto ldB = [&](int kq, int nb) __attribute__((always_inline)) {
        fb[kq][nb] = *(const uint4*)(bsl + bbase16[nb] + (((kq * 4 + g16) ^ bswz16[nb]) * 16));
      };
      ldA(0, 0);
#pragma unroll
      for (int nb = 0; nb < 2 * NT; ++nb) ldB(0, nb);
#pragma unroll
      for (int mb = 1; mb < 2 * MT; ++mb) ldA(mb, 0);
#pragma unroll
      for (int kq = 0; kq < KQ; ++kq)
#pragma unroll
        for (int mb = 0; mb < 2 * MT; ++mb) {
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int nb = 0; nb < 2 * NT; ++nb) Tr<T>::mma16(fa[mb], fb[kq][nb], acc16[mb][nb]);
          if (kq + 1 < KQ) {   // the next k-step's fragments behind this block row's MFMAs: its A row in place, a share of its B set
            ldA(mb, kq + 1);
#pragma unroll
            for (int nb = 0; nb < 2 * NT; ++nb)
              if (nb * 2 * MT / (2 * NT) == mb) ldB(kq + 1, nb);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
    } else {
    constexpr int NFB = KSTEPS >= 3 ? 3 : KSTEPS;   // fragment register sets (a 64-byte K chunk has two k-steps: both fit up front)
    uint4 af[NFB][MT], bf[NFB][NT];
    auto frag_load = [&](int buf, int ks) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) af[buf][mt] = *(const uint4*)(awin + abase[mt] + toff + ks * 32);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bf[buf][nt] = *(const uint4*)(bsl + bbase[nt] + (((ks * 2 + h) ^ bswz[nt]) * 16));
    };
    frag_load(0, 0);
    if (KSTEPS > 1) frag_load(1, 1);
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      if (ks + 2 < KSTEPS) frag_load((ks + 2) % NFB, ks + 2);
      // pin the stage order: left alone, hipcc sinks every read next to its MFMA (lgkmcnt(1) in front of
      // almost every MFMA pair) and the LDS latency is paid k-step by k-step
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) Tr<T>::mma(af[ks % NFB][mt], bf[ks % NFB][nt], acc[mt][nt]);
      __builtin_amdgcn_sched_barrier(0);
    }
    }
  };

  // the same with filler work behind the MFMAs of k-steps 0, 1 and 2 (RB = 128: four k-steps): the MFMAs just
  // issued execute while the wave runs the filler, instead of every wave leaving the pipe idle in a common
  // issue / store phase at the end of the tap
  auto mma_tap_f = [&](const char* awin, const char* bsl, int toff, auto&& f0, auto&& f1, auto&& f2) __attribute__((always_inline)) {
    if constexpr (M16) {
      // A fragments of the current 64-byte k-step (replaced in place behind the last MFMA of their block row), B fragments of both k-steps
      constexpr int KQ = RB / 64;
      uint4 fa[2 * MT], fb[KQ][2 * NT];
      auto ldA = [&](int mb, int kq) __attribute__((always_inline)) { fa[mb] = *(const uint4*)(awin + abase16[mb] + toff + kq * 64); };
      auto ldB = [&](int kq, int nb) __attribute__((always_inline)) {
        fb[kq][nb] = *(const uint4*)(bsl + bbase16[nb] + (((kq * 4 + g16) ^ bswz16[nb]) * 16));
      };
      ldA(0, 0);
#pragma unroll
      for (int nb = 0; nb < 2 * NT; ++nb) ldB(0, nb);
#pragma unroll
      for (int mb = 1; mb < 2 * MT; ++mb) ldA(mb, 0);
#pragma unroll
      for (int kq = 0; kq < KQ; ++kq)
#pragma unroll
        for (int mb = 0; mb < 2 * MT; ++mb) {
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int nb = 0; nb < 2 * NT; ++nb) Tr<T>::mma16(fa[mb], fb[kq][nb], acc16[mb][nb]);
          if (kq + 1 < KQ) {   // the next k-step's fragments behind this block row's MFMAs: its A row in place, a share of its B set
            ldA(mb, kq + 1);
#pragma unroll
            for (int nb = 0; nb < 2 * NT; ++nb)
              if (nb * 2 * MT / (2 * NT) == mb) ldB(kq + 1, nb);
          }
          __builtin_amdgcn_sched_barrier(0);
          {   // the fillers of the 32x32 schedule (behind k-steps 0, 1, 2 of four) keep their places in the MFMA stream
            constexpr int NSTEP = KQ * 2 * MT;
            const int step = kq * 2 * MT + mb;
            if (step == NSTEP / 4 - 1 || (NSTEP < 4 && step == 0)) f0();
            if (step == NSTEP / 2 - 1) f1();
            if (step == (3 * NSTEP) / 4 - 1 || (NSTEP < 4 && step == NSTEP - 1)) f2();
            __builtin_amdgcn_sched_barrier(0);
          }
        }
    } else {
    constexpr int NFB = KSTEPS >= 3 ? 3 : KSTEPS;   // fragment register sets (a 64-byte K chunk has two k-steps: both fit up front)
    uint4 af[NFB][MT], bf[NFB][NT];
    auto frag_load = [&](int buf, int ks) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) af[buf][mt] = *(const uint4*)(awin + abase[mt] + toff + ks * 32);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bf[buf][nt] = *(const uint4*)(bsl + bbase[nt] + (((ks * 2 + h) ^ bswz[nt]) * 16));
    };
    frag_load(0, 0);
    if (KSTEPS > 1) frag_load(1, 1);
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      if (ks + 2 < KSTEPS) frag_load((ks + 2) % NFB, ks + 2);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) Tr<T>::mma(af[ks % NFB][mt], bf[ks % NFB][nt], acc[mt][nt]);
      __builtin_amdgcn_sched_barrier(0);
      if (ks == 0) f0();
      if (ks == 1) f1();
      if (ks == 2 || (KSTEPS == 2 && ks == 1)) f2();   // (64-byte K chunks have two k-steps: the last filler follows the second)
      __builtin_amdgcn_sched_barrier(0);
    }
    }
  };

  // ---------------- prologue: window of chunk 0 + first weight slab ----------------
  {
    Stager cur;
    cur.setup(a.src, a.nsrc, a.Cin, 0, tid);
    if (resident) {
      for (int t = 0; t < a.ntaps; ++t) dmaB(a.tap_w[t] * nchunks, t);
    } else if constexpr (LOOP == LOOP_MASKED || LOOP == LOOP_MASKED_T11) {
      dmaB(a.tap_w[a.src_taps[0] & 15] * nchunks, 0);   // first tap of the first source's list
    } else {
      dmaB(a.tap_w[0] * nchunks, 0);
      if constexpr (LOOP == LOOP_RUN9 || LOOP == LOOP_RUN9S) dmaB(a.tap_w[1] * nchunks, 1);   // three-slot ring: two slabs ahead
    }
    stage_full(cur, ldsA);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#ifdef OCTSEG_STAMP
  STAMP(k1);
#endif

  // ---------------- main loop ----------------
  // PPT = window passes of the NEXT chunk prefetched per tap (1 for multi-tap convs, 4 for 1x1);
  // every load and every LDS store of the pipeline is unconditional: indices are clamped instead
  // (re-staging the last pass / the last chunk again is harmless).
#ifdef OCTSEG_STAMP
  unsigned long long tsum[6] = {0, 0, 0, 0, 0, 0};
#endif
  const int IHl = a.IH, IWl = a.IW, ntaps = a.ntaps;
  // window-pixel cursor of the prefetch: pass p covers pixels p*PSTEP + p0; advancing by one pass is
  // (hy, hx) += (PSTEP / RW, PSTEP % RW) with one carry -- no division in the loop
  const int p0w = tid / VPR;
  const int hy_first = (int)(((float)p0w + 0.5f) * inv_rw), hx_first = p0w - hy_first * RW;
  const int dq = Stager::PSTEP / RW, dr = Stager::PSTEP - dq * RW;
  auto run = [&](auto ppt_c, auto dbuf_c) {
    constexpr int PPT = decltype(ppt_c)::value;
    constexpr bool DBUF = decltype(dbuf_c)::value;
    int it = 0;
    int tap2 = ntaps == 1 ? 0 : 1, chunk2 = ntaps == 1 ? min(1, nchunks - 1) : 0;  // slab of iteration 1
    for (int chunk = 0; chunk < nchunks; ++chunk) {
      const bool has_next = chunk + 1 < nchunks;
      Stager nxt;
      nxt.setup(a.src, a.nsrc, a.Cin, has_next ? chunk + 1 : chunk, tid, a.src_uniform != 0);
      nxt.bind_image(n);
      const char* awin = ldsA + ((DBUF && (chunk & 1)) ? abytes : 0);
      char* anext = ldsA + ((chunk & 1) ? 0 : abytes);
      int hy = hy_first, hx = hx_first, hp = p0w;      // cursor of the pass to prefetch next
      char* wrow = anext + p0w * PITCH;
      for (int t = 0; t < ntaps; ++t, ++it) {
        unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0;
        (void)s0; (void)s1; (void)s2; (void)s3; (void)s4; (void)s5;
        STAMP(s0);
        // window slice of the next chunk (register load) and the LDS-DMA of the next iteration's slab:
        // both fly under the MFMAs below and are retired in front of the barrier.
        uint4 av[PPT];
        bool aok[PPT];
        char* wr[PPT];
        if constexpr (DBUF) {
#pragma unroll
          for (int u = 0; u < PPT; ++u) {
            av[u] = nxt.load_at(hy, hx, hp < npix, gy0, gx0, smul, IHl, IWl, aok[u]);
            wr[u] = wrow;
            // advance to the next pass unless this was the last one (then it is simply staged again)
            const bool adv = hp + Stager::PSTEP < npass * Stager::PSTEP;
            if (adv) {
              hp += Stager::PSTEP; wrow += Stager::PSTEP * PITCH;
              hy += dq; hx += dr;
              if (hx >= RW) { hx -= RW; hy += 1; }
            }
          }
        }
        dmaB(__builtin_amdgcn_readlane(v_tapw, tap2) * nchunks + chunk2, (it + 1) & 1);
        const int toff = __builtin_amdgcn_readlane(v_toff, t);
        STAMP(s1);
        mma_tap(awin, ldsB + (it & 1) * BBYTES, toff);
        STAMP(s2);
        // keep the consumers of the prefetched registers (BN affine, LDS stores) behind the MFMA block
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (DBUF) {
#pragma unroll
          for (int u = 0; u < PPT; ++u) nxt.write_at(wr[u], av[u], aok[u]);
        }
        STAMP(s3);
        // the slab of iteration it+1 (and this wave's LDS stores) must have landed before anyone passes
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        STAMP(s4);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        STAMP(s5);
#ifdef OCTSEG_STAMP
        tsum[0] += s1 - s0; tsum[1] += s2 - s1; tsum[2] += s3 - s2; tsum[3] += s4 - s3; tsum[4] += s5 - s4; tsum[5] += 1;
#endif
        // (tap, chunk) cursor of the next iteration's slab (clamped at the very end)
        if (++tap2 == ntaps) { tap2 = 0; chunk2 = min(chunk2 + 1, nchunks - 1); }
      }
      if constexpr (!DBUF) {
        if (has_next) {  // window does not fit twice: restage in place (all waves passed the barrier)
          stage_full(nxt, ldsA);
          __syncthreads();
        }
      }
    }
  };
  // LOOP_T11: the rolled loop with ONE window buffer, the next chunk's window (six passes of a 13 x 13 window) loaded into registers one pass
  // per tap and stored in place behind the chunk's last barrier -- the loads' latency sits under the chunk's MFMAs instead of in front of the next
  auto run_t11 = [&](auto np_c) {
    constexpr int NP = decltype(np_c)::value;   // window passes (host: npass <= NP <= ntaps)
    int it = 0;
    int tap2 = 1, chunk2 = 0;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
      const bool has_next = chunk + 1 < nchunks;
      Stager nxt;
      nxt.setup(a.src, a.nsrc, a.Cin, has_next ? chunk + 1 : chunk, tid, a.src_uniform != 0);
      nxt.bind_image(n);
      uint4 av[NP];
      bool aok[NP];
      int hy = hy_first, hx = hx_first, hp = p0w;
#pragma unroll
      for (int t = 0; t < 9; ++t, ++it) {
        if (t < NP) {
          av[t] = nxt.load_at(hy, hx, hp < npix, gy0, gx0, smul, IHl, IWl, aok[t]);
          const bool adv = hp + Stager::PSTEP < npass * Stager::PSTEP;
          if (adv) {
            hp += Stager::PSTEP;
            hy += dq; hx += dr;
            if (hx >= RW) { hx -= RW; hy += 1; }
          }
        }
        dmaB(__builtin_amdgcn_readlane(v_tapw, tap2) * nchunks + chunk2, (it + 1) & 1);
        const int toff = __builtin_amdgcn_readlane(v_toff, t);
        mma_tap(ldsA, ldsB + (it & 1) * BBYTES, toff);
        __builtin_amdgcn_sched_barrier(0);
        // the slab of iteration it + 1 must have landed before anyone passes; the window loads may stay in flight (they are older than
        // the DMA only in the taps that issued one: wait for everything but keep it simple -- hipcc does not know the DMA)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (++tap2 == 9) { tap2 = 0; chunk2 = min(chunk2 + 1, nchunks - 1); }
      }
      if (has_next) {   // every wave is past the last tap's barrier: the window can be replaced
#pragma unroll
        for (int u = 0; u < NP; ++u)
          if (u < npass) nxt.write_at(ldsA + (p0w + u * Stager::PSTEP) * PITCH, av[u], aok[u]);
        __syncthreads();
      }
    }
  };
  // Per-source tap subsets (ConvArgs::taps_per_src): `run` with the tap loop walking the list of the source that owns the chunk -- four taps
  // per 64-channel chunk instead of nine, so the next chunk's window (six passes) is prefetched two passes per tap.  Host-checked: double-
  // buffered window, sources on chunk boundaries, npass <= 2 * taps_per_src.
  auto run_masked = [&](auto ppt_c) {
    constexpr int PPT = decltype(ppt_c)::value;
    const int tps = a.taps_per_src;
    auto list_of = [&](int chunk) -> int {
      int l = a.src_taps[0];
#pragma unroll
      for (int i = 1; i < MAX_SRC; ++i)
        if (i < a.nsrc && chunk * KC >= a.src[i].c0) l = a.src_taps[i];
      return l;
    };
    int it = 0;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
      const bool has_next = chunk + 1 < nchunks;
      const int cnext = has_next ? chunk + 1 : chunk;
      Stager nxt;
      nxt.setup(a.src, a.nsrc, a.Cin, cnext, tid, a.src_uniform != 0);
      nxt.bind_image(n);
      const char* awin = ldsA + ((chunk & 1) ? abytes : 0);
      char* anext = ldsA + ((chunk & 1) ? 0 : abytes);
      const int lst = list_of(chunk), lst_n = list_of(cnext);
      int hy = hy_first, hx = hx_first, hp = p0w;
      char* wrow = anext + p0w * PITCH;
      for (int k = 0; k < tps; ++k, ++it) {
        const int t = (lst >> (4 * k)) & 15;
        uint4 av[PPT];
        bool aok[PPT];
        char* wr[PPT];
        // slab of the next iteration: the next tap of this chunk's list, or the first of the next chunk's
        const bool same = k + 1 < tps;
        const int tn = same ? (lst >> (4 * (k + 1))) & 15 : lst_n & 15;
        const int toff = __builtin_amdgcn_readlane(v_toff, t);
        // the prefetches behind the first k-step's MFMAs, the LDS stores of the window passes behind the third's (mma_tap_f): the MFMAs just
        // issued execute while the wave runs them
        mma_tap_f(awin, ldsB + (it & 1) * BBYTES, toff,
          [&]() {
            dmaB(__builtin_amdgcn_readlane(v_tapw, tn) * nchunks + (same ? chunk : cnext), (it + 1) & 1);
#pragma unroll
            for (int u = 0; u < PPT; ++u) {
              av[u] = nxt.load_at(hy, hx, hp < npix, gy0, gx0, smul, IHl, IWl, aok[u]);
              wr[u] = wrow;
              const bool adv = hp + Stager::PSTEP < npass * Stager::PSTEP;
              if (adv) {
                hp += Stager::PSTEP; wrow += Stager::PSTEP * PITCH;
                hy += dq; hx += dr;
                if (hx >= RW) { hx -= RW; hy += 1; }
              }
            }
          },
          [&]() {},
          [&]() {
#pragma unroll
            for (int u = 0; u < PPT; ++u) nxt.write_at(wr[u], av[u], aok[u]);
          });
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
    }
  };
  // 1x1 convs (one tap, four window passes per K chunk): the rolled loop loads a chunk's window at the top of an
  // iteration and stores it at the bottom of the same one.  Here the loads of chunk c + 2 are issued in iteration c
  // and stored in iteration c + 1 -- a whole iteration of MFMAs later -- with two named register sets (window vectors
  // and the lazy-BN parameters, which follow the chunk) and two stagers, written out for a pair of iterations so
  // that no register copy is needed.  Every load is an asm statement outside hipcc's waitcnt model (as plain loads
  // the compiler reused their destinations for the next addresses and drained vmcnt(0) once per iteration for the
  // write-after-read): per iteration the slab DMA is issued first, then the NLD loads; the stores of chunk c + 1
  // wait for `vmcnt(all this iteration issued)` and name their registers "+v"; `vmcnt(NLD)` in front of the barrier
  // retires the DMA and leaves the young loads in flight.  tools/audit_asm_loads.py checks the register rule.
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
  auto run1p = [&]() __attribute__((always_inline)) {
    constexpr int PPT = 4;
    constexpr int NQ = 2 * (Stager::VEC / 4);                 // parameter vectors: scale, shift (x2 for 8 channels)
    constexpr int NLD = PPT + NQ;                             // asm loads per chunk
    constexpr int NDM = NDMA % NWAVES == 0 ? DPW : 0;         // DMAs EVERY wave issues per iteration (lower bound)
    Stager sA, sB;
    u32x4_t avA0 = {0, 0, 0, 0}, avA1 = avA0, avA2 = avA0, avA3 = avA0, qA0 = avA0, qA1 = avA0, qA2 = avA0, qA3 = avA0;
    u32x4_t avB0 = avA0, avB1 = avA0, avB2 = avA0, avB3 = avA0, qB0 = avA0, qB1 = avA0, qB2 = avA0, qB3 = avA0;
    bool okA[PPT], okB[PPT];
    // the image coordinates of this thread's PPT window pixels do not depend on the chunk: resolved once
    int pyc[PPT], pxc[PPT];
    bool pok[PPT];
    {
      int hy = hy_first, hx = hx_first, hp = p0w;
#pragma unroll
      for (int u = 0; u < PPT; ++u) {
        const int iy = gy0 + hy * smul, ix = gx0 + hx * smul;
        pok[u] = hp < npix && (unsigned)iy < (unsigned)IHl && (unsigned)ix < (unsigned)IWl;
        pyc[u] = min(max(iy, 0), IHl - 1); pxc[u] = min(max(ix, 0), IWl - 1);
        const bool adv = hp + Stager::PSTEP < npass * Stager::PSTEP;
        if (adv) {
          hp += Stager::PSTEP;
          hy += dq; hx += dr;
          if (hx >= RW) { hx -= RW; hy += 1; }
        }
      }
    }
    const bool usrc = a.src_uniform != 0;
#define OCTSEG_LD4(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr))
#define OCTSEG_LD4_16(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(dst) : "v"(ptr))
    auto load_chunk = [&](Stager& sg, int chunk, u32x4_t& v0, u32x4_t& v1, u32x4_t& v2, u32x4_t& v3, u32x4_t& q0, u32x4_t& q1,
                          u32x4_t& q2, u32x4_t& q3, bool (&ok)[PPT]) __attribute__((always_inline)) {
      sg.select(a.src, a.nsrc, a.Cin, min(chunk, nchunks - 1), tid, usrc);
      sg.bind_image(n);
      const char* scp = sg.has_aff ? (const char*)(sg.s.scale + sg.s.cl) : Wp;   // (any readable 32 bytes)
      const char* shp = sg.has_aff ? (const char*)(sg.s.shift + sg.s.cl) : Wp;
      OCTSEG_LD4(q0, scp);
      if constexpr (Stager::VEC == 8) OCTSEG_LD4_16(q1, scp);
      OCTSEG_LD4(q2, shp);
      if constexpr (Stager::VEC == 8) OCTSEG_LD4_16(q3, shp);
      const char* g0 = sg.addr_xy(pyc[0], pxc[0]); OCTSEG_LD4(v0, g0);
      const char* g1 = sg.addr_xy(pyc[1], pxc[1]); OCTSEG_LD4(v1, g1);
      const char* g2 = sg.addr_xy(pyc[2], pxc[2]); OCTSEG_LD4(v2, g2);
      const char* g3 = sg.addr_xy(pyc[3], pxc[3]); OCTSEG_LD4(v3, g3);
#pragma unroll
      for (int u = 0; u < PPT; ++u) ok[u] = sg.cvalid && pok[u];
    };
    auto body = [&](int c, Stager& sld, u32x4_t& l0, u32x4_t& l1, u32x4_t& l2, u32x4_t& l3, u32x4_t& lq0, u32x4_t& lq1, u32x4_t& lq2,
                    u32x4_t& lq3, bool (&okl)[PPT], Stager& sst, u32x4_t& s0, u32x4_t& s1, u32x4_t& s2, u32x4_t& s3, u32x4_t& sq0,
                    u32x4_t& sq1, u32x4_t& sq2, u32x4_t& sq3, const bool (&oks)[PPT]) __attribute__((always_inline)) {
      char* wnext = ldsA + ((c & 1) ? 0 : abytes);
      auto row = [&](int u) { return wnext + (min(u, npass - 1) * Stager::PSTEP + p0w) * PITCH; };
      mma_tap_f(ldsA + ((c & 1) ? abytes : 0), ldsB + (c & 1) * BBYTES, 0,
        [&]() {   // behind k-step 0: slab of the next iteration (oldest in the queue), then the loads of chunk c + 2
          dmaB(a.tap_w[0] * nchunks + min(c + 1, nchunks - 1), (c + 1) & 1);
          load_chunk(sld, c + 2, l0, l1, l2, l3, lq0, lq1, lq2, lq3, okl);
        },
        [&]() {   // behind k-steps 1 and 2: lazy BN + LDS stores of chunk c + 1 (everything the previous iteration issued has landed)
          asm volatile("s_waitcnt vmcnt(%8)" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(sq0), "+v"(sq1), "+v"(sq2), "+v"(sq3)
                       : "n"(NDM + NLD) : "memory");
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            sst.sc[i] = __uint_as_float(sq0[i]); sst.sh[i] = __uint_as_float(sq2[i]);
            if constexpr (Stager::VEC == 8) { sst.sc[4 + i] = __uint_as_float(sq1[i]); sst.sh[4 + i] = __uint_as_float(sq3[i]); }
          }
          sst.write_at(row(0), __builtin_bit_cast(uint4, s0), oks[0]);
          sst.write_at(row(1), __builtin_bit_cast(uint4, s1), oks[1]);
        },
        [&]() {
          sst.write_at(row(2), __builtin_bit_cast(uint4, s2), oks[2]);
          sst.write_at(row(3), __builtin_bit_cast(uint4, s3), oks[3]);
        });
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NLD) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    };
    load_chunk(sB, 1, avB0, avB1, avB2, avB3, qB0, qB1, qB2, qB3, okB);
    int c = 0;
    for (; c + 1 < nchunks; c += 2) {
      body(c, sA, avA0, avA1, avA2, avA3, qA0, qA1, qA2, qA3, okA, sB, avB0, avB1, avB2, avB3, qB0, qB1, qB2, qB3, okB);
      body(c + 1, sB, avB0, avB1, avB2, avB3, qB0, qB1, qB2, qB3, okB, sA, avA0, avA1, avA2, avA3, qA0, qA1, qA2, qA3, okA);
    }
    if (c < nchunks) body(c, sA, avA0, avA1, avA2, avA3, qA0, qA1, qA2, qA3, okA, sB, avB0, avB1, avB2, avB3, qB0, qB1, qB2, qB3, okB);
    // the (clamped, unused) loads of the last iterations are still in flight: their registers stay named until they land
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(avA0), "+v"(avA1), "+v"(avA2), "+v"(avA3), "+v"(qA0), "+v"(qA1), "+v"(qA2), "+v"(qA3),
                 "+v"(avB0), "+v"(avB1), "+v"(avB2), "+v"(avB3), "+v"(qB0), "+v"(qB1), "+v"(qB2), "+v"(qB3) :: "memory");
#undef OCTSEG_LD4
#undef OCTSEG_LD4_16
  };
  // 3x3 taps, double-buffered window, THREE-slot slab ring (host: only when LDS allows): the chunk body is written
  // out tap by tap.  The window slice loaded in tap t is stored to LDS at the end of tap t + 1 -- a whole tap of
  // MFMAs later (stamps: waiting for it in the same tap cost ~800 of a tap's ~2900 cycles) -- and the slabs are
  // fetched two taps ahead.  Every prefetch is outside hipcc's waitcnt bookkeeping: per tap the slab DMA is issued
  // first, then the slice, and the only wait of a tap is `vmcnt(what this tap issued)`: everything the previous tap
  // issued has landed -- the next tap's slab and the slice stored now.  The slice load is an asm global_load
  // (hipcc does not know the DMAs and would wait for the younger ones too); its destination is named "+v" in the
  // wait statement, so no consumer can be scheduled in front of the wait, and the two register sets are separate
  // variables of fully unrolled code.  tools/audit_asm_loads.py checks in the ISA that no instruction touches a
  // destination between its load and its wait ('asm load' rule of cdna_hip_programming.md); run it after any edit.
  auto run9r = [&]() __attribute__((always_inline)) {
    constexpr int D = 3;
    int tap2 = D - 1, chunk2 = 0;                   // slab cursor: two iterations ahead
    int slotC = 0, slotS = D - 1;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
      const bool has_next = chunk + 1 < nchunks;
      Stager nxt;
      nxt.select(a.src, a.nsrc, a.Cin, has_next ? chunk + 1 : chunk, tid, a.src_uniform != 0);
      nxt.bind_image(n);
      // Lazy-BN parameters of that chunk: asm loads like the slices.  As plain loads hipcc waited for them at their
      // first use, which lies inside the exec-masked affine of a tap; at the join behind it its model still held them
      // pending, so EVERY tap's affine got an `s_waitcnt vmcnt(0)` -- which also drained the slab DMAs and the slice
      // that tap had just issued (each tap then paid its own prefetch latency).  Issued here they are older than tap
      // 0's prefetches, so tap 0's counted wait retires them; it names them "+v", and the first use is in tap 1.
      u32x4_t q0 = {0, 0, 0, 0}, q1 = {0, 0, 0, 0}, q2 = {0, 0, 0, 0}, q3 = {0, 0, 0, 0};
      {
        const char* scp = nxt.has_aff ? (const char*)(nxt.s.scale + nxt.s.cl) : Wp;   // (any readable 32 bytes)
        const char* shp = nxt.has_aff ? (const char*)(nxt.s.shift + nxt.s.cl) : Wp;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(q0) : "v"(scp));
        if constexpr (Stager::VEC == 8) asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(q1) : "v"(scp));
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(q2) : "v"(shp));
        if constexpr (Stager::VEC == 8) asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(q3) : "v"(shp));
      }
      const char* awin = ldsA + ((chunk & 1) ? abytes : 0);
      char* anext = ldsA + ((chunk & 1) ? 0 : abytes);
      int hy = hy_first, hx = hx_first, hp = p0w;
      char* wrow = anext + p0w * PITCH;
      u32x4_t avA = {0, 0, 0, 0}, avB = {0, 0, 0, 0};
      bool okA = false, okB = false;
      char* wrA = wrow; char* wrB = wrow;
      auto tap = [&](auto tc) __attribute__((always_inline)) {
        constexpr int TT = decltype(tc)::value;
        constexpr int W = TT < 8 ? 1 : 0;            // a slice is issued in this tap
        unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0;
        (void)s0; (void)s1; (void)s2; (void)s3; (void)s4; (void)s5;
        STAMP(s0);
        const int toff = __builtin_amdgcn_readlane(v_toff, TT);
        const char* bsl = ldsB + slotC * BBYTES;
        if (++slotC == D) slotC = 0;
        uint4 wv = make_uint4(0, 0, 0, 0);
        STAMP(s1);
        mma_tap_f(awin, bsl, toff,
          [&]() {   // behind k-step 0: this tap's prefetches (slab of iteration it + 2 first, then the slice)
            dmaB(__builtin_amdgcn_readlane(v_tapw, tap2) * nchunks + chunk2, slotS);
            if (++slotS == D) slotS = 0;
            if constexpr (TT < 8) {
              bool& ok = (TT & 1) ? okB : okA;
              char*& wr = (TT & 1) ? wrB : wrA;
              const char* g = nxt.addr_at(hy, hx, hp < npix, gy0, gx0, smul, IHl, IWl, ok);
              if constexpr (TT & 1) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(avB) : "v"(g));
              else asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(avA) : "v"(g));
              wr = wrow;
              const bool adv = hp + Stager::PSTEP < npass * Stager::PSTEP;
              if (adv) {
                hp += Stager::PSTEP; wrow += Stager::PSTEP * PITCH;
                hy += dq; hx += dr;
                if (hx >= RW) { hx -= RW; hy += 1; }
              }
            }
          },
          [&]() {   // behind k-step 1: everything the previous tap issued has landed (this tap's DPW + W operations
                    // may fly): the slab of the next tap, and the slice loaded one tap ago -> lazy BN / ReLU on it
            STAMP(s2);
            if constexpr (TT >= 1) {
              if constexpr ((TT - 1) & 1) {
                asm volatile("s_waitcnt vmcnt(%1)" : "+v"(avB) : "n"(DPW + W) : "memory");
                STAMP(s3);
                wv = nxt.prep(__builtin_bit_cast(uint4, avB), okB);
              } else {
                asm volatile("s_waitcnt vmcnt(%1)" : "+v"(avA) : "n"(DPW + W) : "memory");
                STAMP(s3);
                wv = nxt.prep(__builtin_bit_cast(uint4, avA), okA);
              }
            } else {
              asm volatile("s_waitcnt vmcnt(%4)" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "n"(DPW + W) : "memory");
              STAMP(s3);
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                nxt.sc[i] = __uint_as_float(q0[i]); nxt.sh[i] = __uint_as_float(q2[i]);
                if constexpr (Stager::VEC == 8) { nxt.sc[4 + i] = __uint_as_float(q1[i]); nxt.sh[4 + i] = __uint_as_float(q3[i]); }
              }
            }
#ifdef OCTSEG_STAMP
            tsum[2] += s3 - s2;
#endif
          },
          [&]() {   // behind k-step 2: its LDS store
            if constexpr (TT >= 1) *(uint4*)((((TT - 1) & 1) ? wrB : wrA) + nxt.cv * 16) = wv;
          });
        __builtin_amdgcn_sched_barrier(0);
        unsigned long long s2e = 0; (void)s2e;
        STAMP(s2e);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        STAMP(s4);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        STAMP(s5);
#ifdef OCTSEG_STAMP
        tsum[0] += s1 - s0; tsum[1] += s2e - s1; tsum[3] += s4 - s2e; tsum[4] += s5 - s4; tsum[5] += 1;
#endif
        if (++tap2 == 9) { tap2 = 0; chunk2 = min(chunk2 + 1, nchunks - 1); }
      };
      tap(std::integral_constant<int, 0>{}); tap(std::integral_constant<int, 1>{}); tap(std::integral_constant<int, 2>{});
      tap(std::integral_constant<int, 3>{}); tap(std::integral_constant<int, 4>{}); tap(std::integral_constant<int, 5>{});
      tap(std::integral_constant<int, 6>{}); tap(std::integral_constant<int, 7>{}); tap(std::integral_constant<int, 8>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped tail slabs
    __syncthreads();
  };
  // Per-source tap subsets on the 4-wave tile with ONE window buffer (two workgroups per CU): four taps per 64-channel chunk, the next chunk's
  // window (six passes of a 10 x 18 or 13 x 13 window) loaded into registers over those taps (2 + 2 + 1 + 1) and stored in place behind the
  // chunk's last barrier.  Host-checked: taps_per_src == 4, npass <= 6.
  auto run_masked_rp = [&]() {
    constexpr int NP = 6, TPS = 4;
    auto list_of = [&](int chunk) -> int {
      if (a.taps_per_src == 0) return 0x3210;   // a plain four-tap launch (one ConvTranspose2d parity): every chunk meets taps 0 .. 3
      int l = a.src_taps[0];
#pragma unroll
      for (int i = 1; i < MAX_SRC; ++i)
        if (i < a.nsrc && chunk * KC >= a.src[i].c0) l = a.src_taps[i];
      return l;
    };
    int it = 0;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
      const bool has_next = chunk + 1 < nchunks;
      const int cnext = has_next ? chunk + 1 : chunk;
      Stager nxt;
      nxt.setup(a.src, a.nsrc, a.Cin, cnext, tid, a.src_uniform != 0);
      nxt.bind_image(n);
      const int lst = list_of(chunk), lst_n = list_of(cnext);
      uint4 av[NP];
      bool aok[NP];
      int hy = hy_first, hx = hx_first, hp = p0w;
      auto load_pass = [&](auto uc) __attribute__((always_inline)) {
        constexpr int U = decltype(uc)::value;
        av[U] = nxt.load_at(hy, hx, hp < npix, gy0, gx0, smul, IHl, IWl, aok[U]);
        const bool adv = hp + Stager::PSTEP < npass * Stager::PSTEP;
        if (adv) {
          hp += Stager::PSTEP;
          hy += dq; hx += dr;
          if (hx >= RW) { hx -= RW; hy += 1; }
        }
      };
      auto tap = [&](auto kc) __attribute__((always_inline)) {
        constexpr int K = decltype(kc)::value;
        const int t = (lst >> (4 * K)) & 15;
        if constexpr (K == 0) { load_pass(std::integral_constant<int, 0>{}); load_pass(std::integral_constant<int, 1>{}); }
        if constexpr (K == 1) { load_pass(std::integral_constant<int, 2>{}); load_pass(std::integral_constant<int, 3>{}); }
        if constexpr (K == 2) load_pass(std::integral_constant<int, 4>{});
        if constexpr (K == 3) load_pass(std::integral_constant<int, 5>{});
        const bool same = K + 1 < TPS;
        const int tn = same ? (lst >> (4 * (K + 1))) & 15 : lst_n & 15;
        dmaB(__builtin_amdgcn_readlane(v_tapw, tn) * nchunks + (same ? chunk : cnext), (it + 1) & 1);
        const int toff = __builtin_amdgcn_readlane(v_toff, t);
        mma_tap(ldsA, ldsB + (it & 1) * BBYTES, toff);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        ++it;
      };
      tap(std::integral_constant<int, 0>{}); tap(std::integral_constant<int, 1>{});
      tap(std::integral_constant<int, 2>{}); tap(std::integral_constant<int, 3>{});
      if (has_next) {
#pragma unroll
        for (int u = 0; u < NP; ++u)
          if (u < npass) nxt.write_at(ldsA + (p0w + u * Stager::PSTEP) * PITCH, av[u], aok[u]);
        __syncthreads();
      }
    }
  };
  if constexpr (LOOP == LOOP_RESIDENT) {
    for (int t = 0; t < ntaps; ++t) mma_tap(ldsA, ldsB + t * BBYTES, __builtin_amdgcn_readlane(v_toff, t));
    __syncthreads();   // the epilogue reuses the LDS
  } else if constexpr (LOOP == LOOP_RUN9S) {   // host-checked: ONE K chunk, 9 taps, slab pieces divide over the waves
    // single window, nothing to prefetch but the slabs: three-slot ring, the DMA of iteration t + 2 behind the
    // MFMAs of k-step 0, one counted wait per tap (the rolled loop drains vmcnt(0) every tap: these are the K-thin,
    // N-wide data gradients of the decoder's concat layers)
    if constexpr (NDMA % NWAVES == 0 && RB == 128) {
      int slotC = 0, slotS = 2;
      auto tap = [&](auto tc) __attribute__((always_inline)) {
        constexpr int TT = decltype(tc)::value;
        const int toff = __builtin_amdgcn_readlane(v_toff, TT);
        const char* bsl = ldsB + slotC * BBYTES;
        if (++slotC == 3) slotC = 0;
        mma_tap_f(ldsA, bsl, toff,
          [&]() {
            if constexpr (TT + 2 < 9) dmaB(__builtin_amdgcn_readlane(v_tapw, TT + 2) * nchunks, slotS);
            if (++slotS == 3) slotS = 0;
          },
          [&]() {   // the slab of the next tap (issued one tap ago) has landed; this tap's DMA may fly
            if constexpr (TT + 2 < 9) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          },
          [&]() {});
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      };
      tap(std::integral_constant<int, 0>{}); tap(std::integral_constant<int, 1>{}); tap(std::integral_constant<int, 2>{});
      tap(std::integral_constant<int, 3>{}); tap(std::integral_constant<int, 4>{}); tap(std::integral_constant<int, 5>{});
      tap(std::integral_constant<int, 6>{}); tap(std::integral_constant<int, 7>{}); tap(std::integral_constant<int, 8>{});
    }
  } else if constexpr (LOOP == LOOP_RUN9) {   // host-checked: dbuf, 9 taps, npass <= 8, slab pieces divide over the waves
    if constexpr (NDMA % NWAVES == 0 && (RB == 128 || (RB == 64 && NT == 4))) run9r();
  } else if constexpr (LOOP == LOOP_MASKED_T11) {   // host-checked: per-source lists of four taps, one window buffer, six passes
    run_masked_rp();
  } else if constexpr (LOOP == LOOP_MASKED) {   // host-checked: per-source tap lists; dbuf (npass <= 2 * taps_per_src), or one buffer on the 4-wave tile
    if constexpr (WM == 2) { if (dbuf) run_masked(std::integral_constant<int, 2>{}); else run_masked_rp(); }
    else run_masked(std::integral_constant<int, 2>{});
  } else if constexpr (LOOP == LOOP_1X1) {    // host-checked: dbuf, one tap (at most four window passes)
    run1p();   // (the rolled loop is not kept as an A/B switch here: with both in one function hipcc moved the by-value ConvArgs to scratch)
  } else if constexpr (LOOP == LOOP_T11) {   // host-checked: nine taps, one window buffer, six window passes
    run_t11(std::integral_constant<int, 6>{});
  } else {
    if (dbuf) run(std::integral_constant<int, 1>{}, std::true_type{});
    else run(std::integral_constant<int, 1>{}, std::false_type{});
  }

#ifdef OCTSEG_STAMP
  STAMP(k2);
#endif
  conv_epilogue<T, NT, WN, WM, ROWMAP_GROUPED, MT, T11>(a, smem, acc, acc16, tp);
#ifdef OCTSEG_STAMP
  STAMP(k3);
  if (a.stamp != nullptr && lane == 0) {
    for (int i = 0; i < 6; ++i) atomicAdd(a.stamp + i, tsum[i]);
    atomicAdd(a.stamp + 8, k1 - k0); atomicAdd(a.stamp + 9, k2 - k1); atomicAdd(a.stamp + 10, k3 - k2); atomicAdd(a.stamp + 11, 1ull);
  }
#endif
}

// ---------------------------------------------------------------------------------------------
namespace {

struct Variant { int NT, WN, WM, RB; };

size_t variant_lds(const ConvArgs& a, const Variant& v, int esz, int dbuf, int* npass_out) {
  const int TH = 2 * wave_mt(v.NT, v.RB) * v.WM, BN = v.NT * 32 * v.WN, BM = TH * TW, PITCH = v.RB + 16;
  (void)PITCH;
  const bool single = a.ntaps == 1;
  const int RH = single ? TH : (TH - 1) * a.istride + a.span_y;
  const int RW = single ? TW : (TW - 1) * a.istride + a.span_x;
  const int npix = RH * RW;
  const int nthreads = 64 * v.WM * v.WN;
  const int pstep = nthreads / (v.RB / 16);
  if (npass_out) *npass_out = (npix + pstep - 1) / pstep;
  const int npass = (npix + pstep - 1) / pstep;
  const size_t abytes = (size_t)npass * pstep * PITCH;
  const size_t main_loop = (dbuf ? 2 : 1) * abytes + 2 * (size_t)BN * v.RB;
  const size_t epi = (size_t)BM * (BN * esz + 16) + (size_t)v.WM * BN * 2 * sizeof(float);
  return main_loop > epi ? main_loop : epi;
}

template <typename T, int NT, int WN, int WM, int RB, int LOOP>
hipError_t launch_loop(const ConvArgs& a, int mode, size_t lds, hipStream_t st) {
  constexpr int BN = NT * 32 * WN, TH = 2 * wave_mt(NT, RB) * WM;
  const int mtiles = (LOOP == LOOP_T11 || LOOP == LOOP_MASKED_T11) ? a.N * (a.OH / 11) * (a.OW / 11) : a.N * ((a.OH + TH - 1) / TH) * ((a.OW + TW - 1) / TW);
  dim3 grid(mtiles, (a.Cout + BN - 1) / BN);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_mfma_kernel<T, NT, WN, WM, RB, LOOP>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_mfma_kernel<T, NT, WN, WM, RB, LOOP>), grid, dim3(64 * WM * WN), lds, st, a, mode);
  return hipGetLastError();
}

template <typename T, int NT, int WN, int WM, int RB>
hipError_t launch_variant(const ConvArgs& a, int mode, int loop, size_t lds, hipStream_t st) {
  switch (loop) {
    case LOOP_RESIDENT: return launch_loop<T, NT, WN, WM, RB, LOOP_RESIDENT>(a, mode, lds, st);
    case LOOP_1X1: return launch_loop<T, NT, WN, WM, RB, LOOP_1X1>(a, mode, lds, st);
    case LOOP_RUN9:
      if constexpr ((RB == 128 || (RB == 64 && NT == 4)) && ((NT * 32 * WN * RB / 1024) % (WM * WN)) == 0) return launch_loop<T, NT, WN, WM, RB, LOOP_RUN9>(a, mode, lds, st);
      else return hipErrorInvalidValue;
    case LOOP_RUN9S:
      if constexpr (RB == 128 && ((NT * 32 * WN * RB / 1024) % (WM * WN)) == 0) return launch_loop<T, NT, WN, WM, RB, LOOP_RUN9S>(a, mode, lds, st);
      else return hipErrorInvalidValue;
    case LOOP_MASKED:
      if constexpr (RB == 128 && sizeof(T) == 2 && NT == 2) return launch_loop<T, NT, WN, WM, RB, LOOP_MASKED>(a, mode, lds, st);
      else return hipErrorInvalidValue;
    case LOOP_MASKED_T11:
      if constexpr (RB == 128 && sizeof(T) == 2 && NT == 2 && WN == 2 && WM == 2) return launch_loop<T, NT, WN, WM, RB, LOOP_MASKED_T11>(a, mode, lds, st);
      else return hipErrorInvalidValue;
    case LOOP_T11:
      if constexpr (RB == 128 && sizeof(T) == 2 && NT == 2 && WN == 2 && WM == 2) return launch_loop<T, NT, WN, WM, RB, LOOP_T11>(a, mode, lds, st);
      else return hipErrorInvalidValue;
    default: return launch_loop<T, NT, WN, WM, RB, LOOP_GENERIC>(a, mode, lds, st);
  }
}

// Tile choice: N tile from Cout, K chunk from Cin, M tile (16x16 or 8x16 pixels) from tile utilisation
// and LDS fit (double-buffered window preferred).
struct Choice { Variant v; int dbuf; size_t lds; int resident; int ring3; int ring1; int tile11; };   // ring3: run9r, ring1: run9s (one chunk)

// LOOP_T11's launches: 2-byte 3x3 stride-1 layers with > 64 output channels on maps that are multiples of 11 and not of 16, up to 88 pixels a
// side, whose 16-pixel tiling is a grid of at most 4608 workgroups (U-Net++/resnet101 16 x 704^2, ms per step of the MFMA kernels alone: off 58.74,
// <= 1024: 56.38, <= 2304: 56.09, <= 4608: 55.5, no limit 55.68).  `pool_ok` =
// false: the geometry alone (a pooled destination needs even tiles -- such a launch keeps the 16-pixel tiling but must keep the same weight image)
static bool tile11_geom(const ConvArgs& a, int esz) {
  static const bool off = getenv("OCTSEG_NO_TILE11") != nullptr;   // A/B switch
  if (off || esz != 2 || a.taps_per_src > 0) return false;
  if (a.ntaps != 9 || a.span_x != 3 || a.span_y != 3 || a.istride != 1 || a.ostride != 1 || a.ooy != 0 || a.oox != 0) return false;
  if (a.out_mode == OUT_HEAD_NCHW || a.IH != a.OH || a.IW != a.OW || a.OH % 11 != 0 || a.OW % 11 != 0 || a.OH > 88 || a.OW > 88) return false;
  if (a.OH % 16 == 0 && a.OW % 16 == 0) return false;
  if (a.Cout <= 64 || a.Cin < 64) return false;
  const long long wg16 = (long long)a.N * ((a.OH + 15) / 16) * ((a.OW + 15) / 16) * ((a.Cout + 127) / 128);
  static const long long maxwg = getenv("OCTSEG_TILE11_MAXWG") ? atoll(getenv("OCTSEG_TILE11_MAXWG")) : 4608;   // experiments
  // ... and only where the 11-pixel grid gives every CU a workgroup: on a grid that leaves CUs idle a layer takes as long as ONE workgroup, and this
  // loop restages its window between K chunks with nobody to overlap it (fp16 ensemble, one frame: 147 -> 130 frames/s without this bound)
  static const long long minwg = getenv("OCTSEG_TILE11_MINWG") ? atoll(getenv("OCTSEG_TILE11_MINWG")) : 256;
  const long long wg11 = (long long)a.N * (a.OH / 11) * (a.OW / 11) * ((a.Cout + 127) / 128);
  return wg16 <= maxwg && wg11 >= minwg;
}

static Choice choose(const ConvArgs& a, int esz) {
  Choice c;
  c.ring3 = 0; c.ring1 = 0; c.tile11 = 0;
  const bool t11 = tile11_geom(a, esz);
  if (t11) {
    bool pool = false;
    for (int i = 0; i < a.ndst; ++i) pool = pool || a.dst[i].pool != 0;
    if (!pool) {   // 13 x 13 window = 6 passes of 32 pixels, two slab slots: 60 KB -> two workgroups per CU
      c.v = Variant{2, 2, 2, 128}; c.dbuf = 0; c.resident = 0; c.tile11 = 1;
      const size_t main_loop = (size_t)6 * 32 * 144 + 2 * (size_t)128 * 128;
      const size_t epi = (size_t)128 * (128 * esz + 16) + (size_t)2 * 128 * 2 * sizeof(float);
      c.lds = main_loop > epi ? main_loop : epi;
      return c;
    }
  }
  {
    // plain four-tap stride-1 launches (a ConvTranspose2d forward parity: 2 x 2 taps, output stride 2) with > 64 output channels: the masked loop's
    // 4-wave tile with one window buffer and register prefetch (every chunk meets all four taps); OCTSEG_NO_RP4: the rolled loop (A/B switch)
    static const bool no_rp4 = getenv("OCTSEG_NO_RP4") != nullptr;
    if (!no_rp4 && esz == 2 && a.taps_per_src == 0 && a.ntaps == 4 && a.span_x == 2 && a.span_y == 2 && a.istride == 1 &&
        a.out_mode != OUT_HEAD_NCHW && a.Cout > 64 && a.Cin >= 64) {
      c.v = Variant{2, 2, 2, 128}; c.dbuf = 0; c.resident = 0; c.tile11 = 3;   // (3: the masked loop without per-source lists)
      const size_t main_loop = (size_t)6 * 32 * 144 + 2 * (size_t)128 * 128;
      const size_t epi = (size_t)128 * (128 * esz + 16) + (size_t)2 * 128 * 2 * sizeof(float);
      c.lds = main_loop > epi ? main_loop : epi;
      return c;
    }
  }
  if (a.taps_per_src > 0) {   // masked loop: the 128-channel N tile, 64-channel chunks (conv_masked_eligible checked the rest)
    // four taps per source: the 4-wave tile with one window buffer, two workgroups per CU -- on 11 x 11 pixels where the map is a multiple of 11
    // and not of 16 (the 44^2 / 22^2 low-resolution maps of the deep decoder blocks), else 8 x 16 (OCTSEG_MASKED_DBUF: the double-buffered 16 x 16 form)
    static const bool force_dbuf = getenv("OCTSEG_MASKED_DBUF") != nullptr;   // A/B switch
    if (!force_dbuf && a.taps_per_src == 4 && esz == 2 && a.ntaps == 9 && a.span_x == 3 && a.span_y == 3) {
      c.v = Variant{2, 2, 2, 128}; c.dbuf = 0; c.resident = 0;
      bool pool = false;
      for (int i = 0; i < a.ndst; ++i) pool = pool || a.dst[i].pool != 0;
      c.tile11 = (!pool && a.OH % 11 == 0 && a.OW % 11 == 0 && !(a.OH % 16 == 0 && a.OW % 16 == 0) && a.OH <= 88 && a.OW <= 88) ? 1 : 0;
      const size_t main_loop = (size_t)6 * 32 * 144 + 2 * (size_t)128 * 128;
      const size_t epi = (size_t)128 * (128 * esz + 16) + (size_t)2 * 128 * 2 * sizeof(float);
      c.lds = main_loop > epi ? main_loop : epi;
      return c;
    }
    for (int wm = 4; wm >= 2; wm -= 2) {
      const Variant v{2, 2, wm, 128};
      int npass = 0;
      const size_t lds = variant_lds(a, v, esz, 1, &npass);
      if (lds <= (size_t)160 * 1024 && npass <= 2 * a.taps_per_src) { c.v = v; c.dbuf = 1; c.lds = lds; c.resident = 0; return c; }
    }
    c.v = Variant{2, 2, 2, 128}; c.dbuf = 1; c.resident = 0; c.lds = (size_t)1 << 30;   // (never launched: dispatch refuses)
    return c;
  }
  // Wide-N configuration for the 3x3 layers with >= 256 output channels (54 % of U-Net++/resnet101's FLOPs: x_1_2, x_2_2, x_1_1,
  // x_0_0 and the data gradients of the wide concat layers): 16x16 pixels x 256 channels per workgroup, 4 waves of 128 pixels x
  // 128 channels (256 accumulators each: one wave per SIMD owns the whole register file), 64-byte K chunks.  Against the 128-channel tile the staged window serves twice the
  // output channels -- half the activation loads, lazy-BN arithmetic and LDS stores per MFMA, the input read once instead of
  // once per N tile -- and a wave reads 8 fragments per 16 MFMAs instead of 4 per 4.  LDS: 2 x 30 KiB windows + 3 x 16 KiB slabs.
  {
    static const bool off = getenv("OCTSEG_NO_N256") != nullptr;   // A/B switch
    const int n256 = (a.Cout + 255) / 256;
    const int kc = 64 / esz;
    // Measured per layer (U-Net++/resnet101, 16 x 704^2, profiles/r2_layers_alone.csv): 1024 -> 256 @176^2 forward 2252 -> 2072 us
    // (1129 TFLOP/s), 256 -> 256 @176^2 forward 725 -> 622; but 1536 -> 512 @88^2 +12 % and every wide data gradient (K = 64..512,
    // N = 768..3072) +3..14 % slower: one wave per SIMD has nobody to cover the fragment reads at the head of a tap, and the
    // 88^2 grids lose a round.  So: exactly 256 output channels, K >= 256, a grid of >= 1024 workgroups.
    // ... and not where the persistent kernel takes the layer with the 128-channel tile (16-divisible maps, 2-byte types): on
    // v_mfma_f32_16x16x32 it beats this 32x32x16 tile (same box: decoder forward 14.0 -> 13.6 ms, data gradients 15.5 -> 15.4)
    static const bool no_p3 = getenv("OCTSEG_NO_CONV3X3P") != nullptr || getenv("OCTSEG_KEEP_N256") != nullptr;   // A/B switches
    const bool p3 = !no_p3 && esz == 2 && a.ostride == 1 && a.OH % 16 == 0 && a.OW % 16 == 0 && a.IH == a.OH && a.IW == a.OW;
    const bool fits = !p3 && a.Cout == 256 && a.Cin >= 256 && (long long)a.N * ((a.OH + 15) / 16) * ((a.OW + TW - 1) / TW) >= 1024;
    static const bool all = getenv("OCTSEG_N256_ALL") != nullptr;   // experiments: every >= 256-channel 3x3 layer
    if (!off && !t11 && (fits || all) && a.ntaps == 9 && a.istride == 1 && a.Cout > 128 && (n256 * 256 - a.Cout) * 10 <= a.Cout && a.Cin > kc) {
      const Variant v{4, 2, 2, 64};   // 2 x 2 waves of (4 M sub-tiles x 4 N sub-tiles)
      int npass = 0;
      const size_t lds = variant_lds(a, v, esz, 1, &npass);
      const size_t slab = (size_t)256 * 64;
      const long long wgs = (long long)a.N * ((a.OH + 15) / 16) * ((a.OW + TW - 1) / TW) * n256;
      if (wgs >= 384 && npass <= 8 && lds + slab <= (size_t)160 * 1024) {
        c.v = v; c.dbuf = 1; c.lds = lds + slab; c.resident = 0; c.ring3 = 1;
        return c;
      }
    }
  }
  int NT, WN;
  if (a.Cout > 64) { NT = 2; WN = 2; } else if (a.Cout > 32) { NT = 1; WN = 2; } else { NT = 1; WN = 1; }
  {
    // K-thin layers (one K chunk: the data gradients of the 64-channel decoder layers) whose 128-channel tiling leaves a quarter or more of
    // its last N tile empty take 64-channel N tiles: 192 <- 64 @352^2 (the skip half of x_1_3.conv1's tied data gradient) 1.090 -> 0.925 ms.
    // With several K chunks the lost A reuse costs more than the empty half tile (192 <- 256 @176^2: 0.677 -> 0.727 ms).
    static const bool off = getenv("OCTSEG_NO_BN64_THIN") != nullptr;   // A/B switch
    const int t128 = (a.Cout + 127) / 128;
    if (!off && a.Cout > 64 && a.ntaps == 9 && a.Cin <= 128 / esz && a.Cout % 64 == 0 && (t128 * 128 - a.Cout) * 4 >= t128 * 128) { NT = 1; WN = 2; }
  }
  // Small grids (small per-GPU batches -- strong scaling -- and the deepest stages): a multi-tap layer whose 16x16-pixel x 128-channel
  // tiles number fewer than half the CUs runs at the speed of ONE workgroup's loop (U-Net++/resnet101 at 2 frames per GPU: x_0_0.conv1,
  // 3072 -> 256 @44^2, is 36 workgroups of 27648-deep contractions: 166 TFLOP/s).  A 64-channel N tile and the 8x16-pixel M tile
  // quadruple the workgroup count at the price of slab reuse, which an under-filled chip does not miss.
  bool small_grid = false;
  {
    static const bool off = getenv("OCTSEG_NO_SMALLGRID") != nullptr;   // A/B switch
    const int bn = NT * 32 * WN;
    const long long w16 = (long long)a.N * ((a.OH + 15) / 16) * ((a.OW + TW - 1) / TW) * ((a.Cout + bn - 1) / bn);
    small_grid = !off && a.ntaps > 1 && w16 < 128;
    if (small_grid && a.Cout > 64) { NT = 1; WN = 2; }
  }
  const int kc128 = 128 / esz;
  const int RB = a.Cin <= kc128 / 2 ? 64 : 128;
  auto util = [&](int TH) {
    const double ty = (a.OH + TH - 1) / TH, tx = (a.OW + TW - 1) / TW;
    return (double)a.OH * a.OW / (ty * TH * tx * TW);
  };
  const size_t cap = 160 * 1024;
  int wm_first = util(8) > 1.15 * util(16) ? 2 : 4;
  // 1x1 convs and single-chunk (resident-tap) layers run a handful of iterations per tile: what they need is
  // several small workgroups per CU, so that one's prologue / epilogue hides under another's MFMAs
  // (measured +6..23 % on the ResNet bottleneck shapes); the 8x16 tile costs them no halo worth mentioning
  {
    const int kc = (a.Cin <= (128 / esz) / 2 ? 64 : 128) / esz;
    // (not for Cout <= 32: those workgroups are WM waves only, and two-wave workgroups measured 20-40 % slower
    //  than four-wave ones on the thin full-resolution layers)
    if ((a.ntaps == 1 || a.Cin <= kc) && WN > 1) wm_first = 2;
    // ... except the square 64 -> 64 3x3 layers (one N tile): the 16x16 tile measured 3-10 % faster there; the K-thin
    // dgrads of the wide concat layers (several N tiles) keep the small tile (15 % faster)
    if (a.ntaps == 9 && a.Cin <= kc && a.Cout <= 64 && WN > 1) wm_first = 4;
    // 33..64 output channels over several K chunks (U-Net++ x_k_3 / x_0_2 conv1: 320..896 -> 64): 8x16 tiles, two
    // workgroups per CU, measured 7-15 % faster than one 8-wave workgroup (its waves own a single 32-channel column)
    if (a.ntaps == 9 && a.Cin > kc && WN == 2 && NT == 1) wm_first = 2;
    // a 16x16 grid that fits the chip in one round beats more, smaller tiles (layer4 3x3 at 22x22: 256 workgroups, 25 % faster)
    if (a.ntaps == 9 && a.Cin > kc) {
      const int BN = NT * 32 * WN;
      const long long b16 = (long long)a.N * ((a.OH + 15) / 16) * ((a.OW + TW - 1) / TW) * ((a.Cout + BN - 1) / BN);
      const long long b8 = (long long)a.N * ((a.OH + 7) / 8) * ((a.OW + TW - 1) / TW) * ((a.Cout + BN - 1) / BN);
      if (b16 <= 256 && b8 > 256) wm_first = 4;
      // (tried in round 2: 8x16 tiles for the 1.1-round case of ResNet layer3, 288 -> 576 workgroups: 87 -> 93 us per layer, not kept)
    }
  }
  if (small_grid) wm_first = 2;
  static const int force_wm = getenv("OCTSEG_FORCE_WM") ? atoi(getenv("OCTSEG_FORCE_WM")) : 0;   // A/B switch
  if (force_wm) wm_first = force_wm;
  const int order[2] = {wm_first, wm_first == 4 ? 2 : 4};
  const int nchunks_c = (a.Cin + RB / esz - 1) / (RB / esz);
  // a single K chunk never restages its window: a second buffer would only cost occupancy
  for (int pref_dbuf = nchunks_c > 1 ? 1 : 0; pref_dbuf >= 0; --pref_dbuf)
    for (int k = 0; k < 2; ++k) {
      Variant v{NT, WN, order[k], RB};
      int npass = 0;
      const size_t lds = variant_lds(a, v, esz, pref_dbuf, &npass);
      // the pipeline prefetches 1 pass per tap (4 for 1x1): the next window must fit that budget
      const bool fits_pipe = a.ntaps == 1 ? npass <= 4 : npass <= a.ntaps;
      if (lds <= cap && (!pref_dbuf || fits_pipe)) {
        c.v = v; c.dbuf = pref_dbuf; c.lds = lds; c.resident = 0;
        // 3x3 double-buffered: a third slab slot if LDS allows (slabs are then fetched two taps ahead)
        {
          const size_t slab = (size_t)v.NT * 32 * v.WN * v.RB;
          const int nd = (int)(slab / 1024), nw = v.WM * v.WN;
          static const bool no_ring3 = getenv("OCTSEG_NO_RING3") != nullptr || getenv("OCTSEG_NO_RUN9") != nullptr;   // A/B switches
          if (!no_ring3 && pref_dbuf && a.ntaps == 9 && npass <= 8 && nd % nw == 0 && lds + slab <= cap) { c.ring3 = 1; c.lds = lds + slab; }
        }
        // thin layers: one K chunk and slabs small enough to keep all taps in LDS -> no per-tap DMA wait / barrier
        const size_t slab = (size_t)v.NT * 32 * v.WN * v.RB;
        if (nchunks_c == 1 && a.ntaps > 1 && a.ntaps * slab <= 40 * 1024) {
          const size_t extra = (size_t)(a.ntaps - 2) * slab;
          if (lds + extra <= 64 * 1024) { c.resident = 1; c.lds = lds + extra; }
        }
        // one K chunk, 3x3, slabs too big to stay resident: slab ring with counted waits (run9s)
        {
          const int nd = (int)(slab / 1024), nw = v.WM * v.WN;
          static const bool no_run9s = getenv("OCTSEG_NO_RUN9S") != nullptr || getenv("OCTSEG_NO_RUN9") != nullptr;   // A/B switches
          if (!no_run9s && !c.resident && !pref_dbuf && nchunks_c == 1 && a.ntaps == 9 && v.RB == 128 && nd % nw == 0 && lds + slab <= cap) {
            c.ring1 = 1; c.lds = lds + slab;
          }
        }
        return c;
      }
    }
  c.v = Variant{NT, WN, 2, RB}; c.dbuf = 0; c.resident = 0; c.lds = variant_lds(a, c.v, esz, 0, nullptr);
  return c;
}

template <typename T>
hipError_t dispatch(const ConvArgs& a_in, hipStream_t st) {
  const Choice c = choose(a_in, (int)sizeof(T));
  if (c.lds > 160 * 1024) return hipErrorInvalidValue;
  const Variant& v = c.v;
  ConvArgs a = a_in;
  {
    const int KC = v.RB / (int)sizeof(T);
    a.src_uniform = 1;
    for (int i = 1; i < a.nsrc; ++i)
      if (a.src[i].c0 % KC != 0) a.src_uniform = 0;
    static const bool no_usrc = getenv("OCTSEG_NO_UNIFORM_SRC") != nullptr;   // A/B switch
    if (no_usrc) a.src_uniform = 0;
  }
  const int loop = a.taps_per_src > 0 ? (c.tile11 ? LOOP_MASKED_T11 : LOOP_MASKED) : c.tile11 == 3 ? LOOP_MASKED : c.tile11 ? LOOP_T11
                   : c.resident ? LOOP_RESIDENT : (c.ring3 ? LOOP_RUN9 : (c.ring1 ? LOOP_RUN9S : ((a.ntaps == 1 && c.dbuf) ? LOOP_1X1 : LOOP_GENERIC)));
#define OCTSEG_CASE(NT_, WN_, WM_, RB_)                                             \
  if (v.NT == NT_ && v.WN == WN_ && v.WM == WM_ && v.RB == RB_)                     \
    return launch_variant<T, NT_, WN_, WM_, RB_>(a, c.dbuf, loop, c.lds, st);
  OCTSEG_CASE(4, 2, 2, 64)
  OCTSEG_CASE(2, 2, 4, 128)
  OCTSEG_CASE(2, 2, 2, 128)
  OCTSEG_CASE(1, 2, 4, 128)
  OCTSEG_CASE(1, 2, 2, 128)
  OCTSEG_CASE(1, 1, 4, 128)
  OCTSEG_CASE(1, 1, 2, 128)
  OCTSEG_CASE(2, 2, 4, 64)
  OCTSEG_CASE(2, 2, 2, 64)
  OCTSEG_CASE(1, 2, 4, 64)
  OCTSEG_CASE(1, 2, 2, 64)
  OCTSEG_CASE(1, 1, 4, 64)
  OCTSEG_CASE(1, 1, 2, 64)
#undef OCTSEG_CASE
  return hipErrorInvalidValue;
}

}  // namespace

ConvPackInfo conv_pack_info(const ConvArgs& a, int dtype) {
  const Choice c = choose(a, (int)dtype_size(dtype));
  ConvPackInfo p;
  p.BN = c.v.NT * 32 * c.v.WN;
  p.RB = c.v.RB;
  const int KC = p.RB / (int)dtype_size(dtype);
  p.nchunks = (a.Cin + KC - 1) / KC;
  p.ntiles = (a.Cout + p.BN - 1) / p.BN;
  return p;
}

int conv_num_mtiles(const ConvArgs& a, int dtype) {
  const Choice c = choose(a, (int)dtype_size(dtype));
  if (c.tile11 == 1) return a.N * (a.OH / 11) * (a.OW / 11);
  const int TH = 2 * wave_mt(c.v.NT, c.v.RB) * c.v.WM;
  return a.N * ((a.OH + TH - 1) / TH) * ((a.OW + TW - 1) / TW);
}

// A stride-1 1x1 convolution does not care about image geometry: present the N*H*W pixels as one image of
// 16-pixel rows so that every tile is full (a 22x22 map otherwise fills 47 % of its 16x16 tiles).
static bool flatten_1x1(ConvArgs& a) {
  if (a.ntaps != 1 || a.istride != 1 || a.ostride != 1 || a.out_mode == OUT_HEAD_NCHW) return false;
  if (a.tap_dy[0] != 0 || a.tap_dx[0] != 0) return false;
  const long long npix = (long long)a.N * a.OH * a.OW;
  if (npix % TW != 0 || a.IH != a.OH || a.IW != a.OW) return false;
  for (int i = 0; i < a.nsrc; ++i)
    if (a.src[i].up || a.src[i].H != a.IH || a.src[i].W != a.IW) return false;
  for (int i = 0; i < a.ndst; ++i)
    if (a.dst[i].H != a.OH || a.dst[i].W != a.OW) return false;
  const int rows = (int)(npix / TW);
  a.N = 1; a.IH = a.OH = rows; a.IW = a.OW = TW;
  for (int i = 0; i < a.nsrc; ++i) { a.src[i].H = rows; a.src[i].W = TW; }
  for (int i = 0; i < a.ndst; ++i) { a.dst[i].H = rows; a.dst[i].W = TW; }
  return true;
}

int conv_num_mtiles_flat(const ConvArgs& a0, int dtype) {
  if (thin_conv_eligible(a0, dtype)) return thin_conv_rows(a0);
  if (gemm1x1_eligible(a0, dtype)) return gemm1x1_rows(a0);
  if (conv3x3p_eligible(a0, dtype)) return conv3x3p_rows(a0);
  ConvArgs a = a0;
  flatten_1x1(a);
  return conv_num_mtiles(a, dtype);
}

// what the masked loop needs of a launch with per-source tap subsets (the plan falls back to its other form when this says no)
bool conv_masked_eligible(const ConvArgs& a, int dtype) {
  static const bool off = getenv("OCTSEG_NO_MASKED_LOOP") != nullptr;   // A/B switch
  if (off || dtype == DT_F32 || a.taps_per_src <= 0 || a.taps_per_src > 8 || a.ntaps < 2 || a.ntaps > 16) return false;
  if (a.istride != 1 || a.ostride != 1 || a.out_mode == OUT_HEAD_NCHW || a.Cout <= 64 || a.Cout % 8 != 0) return false;
  const int KC = 64;
  for (int i = 0; i < a.nsrc; ++i)
    if (a.src[i].c0 % KC != 0 || a.src[i].up) return false;
  if (a.Cin % KC != 0) return false;
  const Choice c = choose(a, (int)dtype_size(dtype));
  return c.lds <= (size_t)160 * 1024;
}

hipError_t launch_conv(int dtype, const ConvArgs& a0, hipStream_t st) {
  if (a0.ntaps <= 0) return hipSuccess;
  if (a0.taps_per_src > 0) {
    if (!conv_masked_eligible(a0, dtype)) return hipErrorInvalidValue;
    ConvArgs a = a0;
    a.tile_order = 1;
    return dtype == DT_F16 ? dispatch<f16_t>(a, st) : dispatch<bf16_t>(a, st);
  }
  if (thin_conv_eligible(a0, dtype)) return launch_thin_conv(dtype, a0, st);
  if (gemm1x1_eligible(a0, dtype)) return launch_gemm1x1(dtype, a0, st);
  if (conv3x3p_eligible(a0, dtype)) return launch_conv3x3p(dtype, a0, st);
  ConvArgs a = a0;
  flatten_1x1(a);
  {
    // weight set small enough to stay in every XCD's L2 next to the windows -> M-major order (see map_tile)
    int wt = 0;
    for (int t = 0; t < a.ntaps; ++t) wt = a.tap_w[t] + 1 > wt ? a.tap_w[t] + 1 : wt;
    const size_t wbytes = (size_t)wt * a.Cout * a.Cin * dtype_size(dtype);
    static const int force = getenv("OCTSEG_TILE_ORDER") ? atoi(getenv("OCTSEG_TILE_ORDER")) : -1;   // A/B switch
    (void)wbytes;   // measured: M-major is never slower, also when the weight set exceeds an L2 (dgrads of the wide layers: -3..14 %)
    a.tile_order = force >= 0 ? force : 1;
  }
  if (dtype == DT_F32) return dispatch<float>(a, st);
  if (dtype == DT_F16) return dispatch<f16_t>(a, st);
  return dispatch<bf16_t>(a, st);
}

}  // namespace octseg
