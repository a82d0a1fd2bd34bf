// gemm1x1.hip -- stride-1 1x1 convolutions (ResNet bottleneck conv1 / conv3, LinkNet decoder 1x1s, and their data
// gradients) as a persistent NHWC GEMM on gfx950:  Y[m][n] = sum_k f(X[m][k]) * W[n][k],  m = pixel, f = the lazy
// BatchNorm + ReLU of the source (or identity).
//
// Why a second kernel: through conv_mfma_kernel these layers ran at 10-12 % of the MFMA roof and ~2.3 TB/s (round 1,
// profiles/r1_layers_alone.csv): a 16-iteration K loop per workgroup under a ~25 k-cycle prologue + epilogue, every
// activation staged through LDS, one workgroup per output tile.  They are bandwidth-side GEMMs (AI 100-200 FLOP/B),
// so this kernel is built around the HBM stream:
//   * activations never touch LDS: lane (r, h) of a wave owns pixel row r and loads 64 CONSECUTIVE bytes of it per 64-wide
//     K step (4 x global_load_dwordx4 = the A fragments of four v_mfma_f32_32x32x16 k-steps; the two halves of a wave cover
//     one full 128-byte line per row).  K is a dummy index, so the k-order inside a step is permuted to fit -- weights are
//     addressed with the same permutation.  Three K steps per wave are in flight in registers (12 KiB per wave);
//   * weights: the (K step, N tile) slab of the existing packed image (16 KiB contiguous, XOR-swizzled rows) goes
//     global -> registers -> LDS one step ahead, two LDS slots, one barrier per K step; B fragments are conflict-free
//     ds_read_b128 (the swizzle of pack_all);
//   * persistent: <= 512 workgroups of 4 waves (two per CU), each walking its (M tile) items with the software pipeline
//     running across item boundaries -- the prefetch of the next tile's first K steps overlaps the epilogue of this one;
//   * the epilogue is wave-local (no barrier): BatchNorm partial sums in registers across all tiles of the workgroup
//     (one slab row per workgroup, deterministic), bf16 tile transposed through a private LDS strip, 16-byte stores /
//     accumulates of whole channel vectors; the lazy BN affine of the source runs on the fragments in registers, once
//     per element (each pixel row belongs to one wave).
// Everything is plain HIP (no asm loads / counted waits): hipcc sees every memory operation of the pipeline.
#include "common.h"
#include "conv_common.h"
#include "kernels.h"

#include <cstdlib>
#include <type_traits>

namespace octseg {

constexpr int G1_BM = 128;     // pixel rows per item: 4 waves x 32
constexpr int G1_D = 3;        // K steps of activations in flight per wave
constexpr int G1_MAXWG = 512;  // two workgroups per CU x 256 CUs (gfx950 / MI355X only build)

struct G1Args {
  const char* x; long long xstride;      // source rows: bytes per pixel
  const float* scale; const float* shift; int relu;
  const char* w;                         // packed image [K step][N tile][BN rows][128 B]
  char* y; long long ystride; int accum;
  int M, K, N, n_tiles, m_tiles;
  float* slab; int slab_row0;
  const float* bias; int relu_out;       // eval with folded BatchNorm: y = relu?(acc + bias[n])
};

// BRES: the whole K x BN weight panel of the workgroup's N tile stays in LDS (K <= 256: four slabs, 64 KiB) for all of its M tiles --
// no per-step weight traffic, NO barrier in the main loop (waves run free: one's epilogue overlaps the others' loads and
// MFMAs), one workgroup per CU with the whole register file, six K steps of activations in flight per wave.  For the
// output-heavy layers (N >= 4 tiles: bottleneck conv3, the data gradient of conv1), whose items are only K / 64 <= 4 steps long.
// NEGATIVE RESULT (round 2), kept opt-in: see g1_geom.
template <typename T, int NT, bool AFF, bool BRES>
__global__ __launch_bounds__(256, BRES ? 1 : 2) void gemm1x1_kernel(const G1Args a) {
  constexpr int D = BRES ? 6 : G1_D;            // K steps of activations in flight per wave
  constexpr int NSET = D + 1;                   // register sets of the activation ring
  constexpr int NSLOT = BRES ? 4 : 2;           // weight slabs in LDS
  constexpr int BN = NT * 32;
  constexpr int SLAB = BN * 128;                 // bytes of one (K step, N tile) weight slab
  constexpr int NPB = SLAB / 4096;               // 16-byte pieces of the slab per thread (1, 2 or 4)
  constexpr int OPITCH = BN * 2 + 16;            // transposed-tile pitch of the epilogue strip
  constexpr int PPR = BN / 8;                    // 16-byte pieces per output row
  constexpr int RPI = 64 / PPR;                  // rows stored per wave instruction
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ldsB = smem;                                           // [2][SLAB]
  char* strip0 = smem + NSLOT * SLAB;                          // [4 waves][32][OPITCH]
  float* lds_ss = (float*)(strip0 + 4 * 32 * OPITCH);          // AFF: scale[K], shift[K]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int G = gridDim.x;
  const int wid = blockIdx.x;
  // XCD x (= wid % 8) owns a contiguous range of logical ids: the N tiles of one M tile then share an L2
  const int l = (G & 7) == 0 ? (wid & 7) * (G >> 3) + (wid >> 3) : wid;
  const int nt_idx = l % a.n_tiles, wg_m = l / a.n_tiles, Gm = G / a.n_tiles;
  const int nloc = (a.m_tiles - wg_m + Gm - 1) / Gm;            // M tiles of this workgroup: wg_m, wg_m + Gm, ...
  const int S = a.K >> 6;
  const int total = nloc * S;
  const int n0 = nt_idx * BN;

  if constexpr (AFF) {
    for (int k = tid; k < a.K; k += 256) { lds_ss[k] = a.scale[k]; lds_ss[a.K + k] = a.shift[k]; }
  }

  // ---- activation prefetch cursor (per lane: its pixel row of the tile, its 64-byte half of the K step)
  int pf_s = 0, pf_mt = wg_m;
  auto row_ptr = [&](int mt) -> const char* {
    const int row = min(mt * G1_BM + wave * 32 + r, a.M - 1);
    return a.x + (long long)row * a.xstride + 64 * h;
  };
  const char* pf_ptr = row_ptr(pf_mt);
  uint4 A[NSET][4];
  auto load_A = [&](uint4 (&dst)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) dst[j] = *(const uint4*)(pf_ptr + 16 * j);
    pf_ptr += 128;
    if (++pf_s == S) {   // next item of this workgroup (past the last one: stay there, the loads are harmless)
      pf_s = 0;
      if (pf_mt + Gm < a.m_tiles) pf_mt += Gm;
      pf_ptr = row_ptr(pf_mt);
    }
  };
  // ---- weight slab cursor
  int bs = 0;
  const char* wbase = a.w + (long long)nt_idx * SLAB + tid * 16;
  const long long wstep = (long long)a.n_tiles * SLAB;
  // named registers, not an array: hipcc kept `uint4 Breg[NPB]` (live across the loop back-edge) in scratch memory and waited
  // for the slab loads right behind their issue to spill them
  uint4 B0 = make_uint4(0, 0, 0, 0), B1 = B0, B2 = B0, B3 = B0;
#define G1_LOAD_B()                                                        \
  do {                                                                     \
    const char* p_ = wbase + bs * wstep;                                   \
    B0 = *(const uint4*)(p_);                                              \
    if constexpr (NPB > 1) B1 = *(const uint4*)(p_ + 4096);                \
    if constexpr (NPB > 2) { B2 = *(const uint4*)(p_ + 8192); B3 = *(const uint4*)(p_ + 12288); } \
    if (++bs == S) bs = 0;                                                 \
  } while (0)
#define G1_STORE_B(slot_)                                                  \
  do {                                                                     \
    char* q_ = ldsB + (slot_) * SLAB + tid * 16;                           \
    *(uint4*)(q_) = B0;                                                    \
    if constexpr (NPB > 1) *(uint4*)(q_ + 4096) = B1;                      \
    if constexpr (NPB > 2) { *(uint4*)(q_ + 8192) = B2; *(uint4*)(q_ + 12288) = B3; } \
  } while (0)

  f32x16_t acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[nt][v] = 0.f;
  float s1[NT], s2[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) { s1[nt] = 0.f; s2[nt] = 0.f; }

  float obias[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) obias[nt] = (a.bias != nullptr && n0 + nt * 32 + r < a.N) ? a.bias[n0 + nt * 32 + r] : 0.f;
  const float ofloor = a.relu_out ? 0.f : NO_FLOOR;
  const int bswz = (r >> 1) & 7;
  const int bbase = r * 128;
  const unsigned floor16 = a.relu ? 0u : 0x80008000u;

  // ---- prologue: slab 0 into slot 0, slab 1 in registers (BRES: every slab of the panel into its slot), D activation steps in flight
  if constexpr (BRES) {
    for (int q = 0; q < S; ++q) { G1_LOAD_B(); G1_STORE_B(q); }
  } else {
    G1_LOAD_B();
    G1_STORE_B(0);
    G1_LOAD_B();
  }
#pragma unroll
  for (int d = 0; d < D; ++d) load_A(A[d]);
  __syncthreads();

  int s = 0, mt = wg_m;
  char* strip = strip0 + wave * 32 * OPITCH;

  auto epilogue = [&]() __attribute__((always_inline)) {
    const int m0 = mt * G1_BM + wave * 32;
    if (a.bias != nullptr) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[nt][v] = clamp_lo(acc[nt][v] + obias[nt], ofloor);
    }
    if (a.slab != nullptr) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int v = 0; v < 16; ++v) { const float y = acc[nt][v]; s1[nt] += y; s2[nt] += y * y; }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int row = (v & 3) + 8 * (v >> 2) + 4 * h;
        Tr<T>::store(strip + row * OPITCH, nt * 32 + r, acc[nt][v]);
        acc[nt][v] = 0.f;
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the strip is private to this wave: its own stores have landed
    const int piece = lane % PPR, rsub = lane / PPR;
    const int n = n0 + piece * 8;
    const bool nok = n < a.N;
    char* gp0 = a.y + (long long)(m0 + rsub) * a.ystride + n * 2;
    const long long gstep = (long long)RPI * a.ystride;
    if (a.accum) {   // gradient accumulation: every old vector requested before the first add
      uint4 old[32 / RPI];
#pragma unroll
      for (int i = 0; i < 32 / RPI; ++i) {
        const bool ok = nok && m0 + rsub + i * RPI < a.M;
        old[i] = *(const uint4*)(ok ? gp0 + i * gstep : (char*)a.y);
      }
#pragma unroll
      for (int i = 0; i < 32 / RPI; ++i) {
        const int row = rsub + i * RPI;
        if (nok && m0 + row < a.M) {
          const uint4 val = *(const uint4*)(strip + row * OPITCH + piece * 16);
          float x0[8], x1[8];
          Tr<T>::unpack8(val, x0); Tr<T>::unpack8(old[i], x1);
#pragma unroll
          for (int e = 0; e < 8; ++e) x0[e] += x1[e];
          *(uint4*)(gp0 + i * gstep) = Tr<T>::pack8(x0);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 32 / RPI; ++i) {
        const int row = rsub + i * RPI;
        if (nok && m0 + row < a.M) *(uint4*)(gp0 + i * gstep) = *(const uint4*)(strip + row * OPITCH + piece * 16);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // strip reads done before the next tile's stores reuse it
  };

  auto step = [&](auto set_c, int g) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_c)::value;
    const int slot = BRES ? s : (g & 1);
    if constexpr (!BRES) {
      // weights of step g + 1 (in registers since the previous step) into the other slot, then fetch those of step g + 2
      G1_STORE_B(slot ^ 1);
      G1_LOAD_B();
    }
    // activations of step g + D
    load_A(A[(SET + D) % NSET]);
    const bool tail = (mt * G1_BM + G1_BM > a.M);
    const bool valid = mt * G1_BM + wave * 32 + r < a.M;
    const char* bsl = ldsB + slot * SLAB + bbase;
    uint4 bf[2][NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bf[0][nt] = *(const uint4*)(bsl + nt * 4096 + (((4 * h + 0) ^ bswz) * 16));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j + 1 < 4) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bf[(j + 1) & 1][nt] = *(const uint4*)(bsl + nt * 4096 + (((4 * h + j + 1) ^ bswz) * 16));
      }
      uint4 af = A[SET][j];
      if constexpr (AFF) {
        const int k0 = s * 64 + 32 * h + 8 * j;
        float sc[8], sh[8];
        const float4 c0 = *(const float4*)(lds_ss + k0), c1 = *(const float4*)(lds_ss + k0 + 4);
        const float4 d0 = *(const float4*)(lds_ss + a.K + k0), d1 = *(const float4*)(lds_ss + a.K + k0 + 4);
        sc[0] = c0.x; sc[1] = c0.y; sc[2] = c0.z; sc[3] = c0.w; sc[4] = c1.x; sc[5] = c1.y; sc[6] = c1.z; sc[7] = c1.w;
        sh[0] = d0.x; sh[1] = d0.y; sh[2] = d0.z; sh[3] = d0.w; sh[4] = d1.x; sh[5] = d1.y; sh[6] = d1.z; sh[7] = d1.w;
        af = Tr<T>::affine_floor(af, sc, sh, floor16);
      }
      if (tail && !valid) af = make_uint4(0, 0, 0, 0);   // rows past M contribute nothing (statistics, stores are masked too)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) Tr<T>::mma(af, bf[j & 1][nt], acc[nt]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (++s == S) {
      epilogue();
      s = 0;
      mt += Gm;
    }
    if constexpr (!BRES) __syncthreads();
  };

  for (int g = 0; g < total; g += NSET) {
    step(std::integral_constant<int, 0>{}, g);
    if (g + 1 < total) step(std::integral_constant<int, 1>{}, g + 1);
    if (g + 2 < total) step(std::integral_constant<int, 2>{}, g + 2);
    if (g + 3 < total) step(std::integral_constant<int, 3>{}, g + 3);
    if constexpr (NSET > 4) {
      if (g + 4 < total) step(std::integral_constant<int, 4 % NSET>{}, g + 4);
      if (g + 5 < total) step(std::integral_constant<int, 5 % NSET>{}, g + 5);
      if (g + 6 < total) step(std::integral_constant<int, 6 % NSET>{}, g + 6);
    }
  }
  if constexpr (BRES) __syncthreads();   // the strips double as the reduction buffer below

#undef G1_LOAD_B
#undef G1_STORE_B
  // ---- BatchNorm partial sums of this workgroup: one slab row (every wave holds its rows' share of all BN channels)
  if (a.slab != nullptr) {
    float* red = (float*)strip0;   // [4 waves][BN][2]: the strips are free (barrier at the end of the last step)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      s1[nt] += __shfl_xor(s1[nt], 32);
      s2[nt] += __shfl_xor(s2[nt], 32);
      if (h == 0) { red[(wave * BN + nt * 32 + r) * 2] = s1[nt]; red[(wave * BN + nt * 32 + r) * 2 + 1] = s2[nt]; }
    }
    __syncthreads();
    if (tid < BN && n0 + tid < a.N) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) { t1 += red[(w * BN + tid) * 2]; t2 += red[(w * BN + tid) * 2 + 1]; }
      float* o = a.slab + ((size_t)(a.slab_row0 + wg_m) * a.N + n0 + tid) * 2;
      o[0] = t1; o[1] = t2;
    }
  }
}

namespace {

struct G1Geom { int NT, n_tiles, m_tiles, G, rows, bres; size_t lds; };

static G1Geom g1_geom(const ConvArgs& a) {
  G1Geom g;
  g.NT = a.Cout > 64 ? 4 : (a.Cout > 32 ? 2 : 1);   // = the N tile choose() gives these layers (BN 128 / 64 / 32, RB 128)
  const int BN = g.NT * 32;
  g.n_tiles = (a.Cout + BN - 1) / BN;
  const long long M = (long long)a.N * a.OH * a.OW;
  g.m_tiles = (int)((M + G1_BM - 1) / G1_BM);
  // opt-in (OCTSEG_G1_BRES=1): measured SLOWER than the streaming form on U-Net++/resnet101 (bottleneck conv3 forward 1.97 -> 2.66 ms
  // per step, conv1 data gradients 1.74 -> 2.24): four waves per CU do not cover the HBM latency that eight (two workgroups) do
  static const bool use_bres = getenv("OCTSEG_G1_BRES") != nullptr;
  g.bres = (use_bres && g.NT == 4 && a.Cin <= 256 && a.Cin >= 128 && g.n_tiles >= 4) ? 1 : 0;
  int gm = (g.bres ? G1_MAXWG / 2 : G1_MAXWG) / g.n_tiles;   // BRES: one workgroup per CU
  if (gm < 1) gm = 1;
  if (gm > g.m_tiles) gm = g.m_tiles;
  g.rows = gm;
  g.G = gm * g.n_tiles;
  const bool aff = a.src[0].scale != nullptr;
  g.lds = (size_t)(g.bres ? 4 : 2) * BN * 128 + (size_t)4 * 32 * (BN * 2 + 16) + (aff ? (size_t)a.Cin * 8 : 0);
  return g;
}

}  // namespace

bool gemm1x1_eligible(const ConvArgs& a, int dtype) {
  static const bool off = getenv("OCTSEG_NO_GEMM1X1") != nullptr;   // A/B switch
  if (off || dtype == DT_F32) return false;
  if (a.ntaps != 1 || a.istride != 1 || a.ostride != 1 || a.out_mode == OUT_HEAD_NCHW) return false;
  if (a.tap_dy[0] != 0 || a.tap_dx[0] != 0 || a.nsrc != 1 || a.ndst != 1) return false;
  if (a.bias != nullptr && a.stat_slab != nullptr) return false;   // (bias only as the folded BatchNorm shift of eval forwards)
  const SrcDesc& s = a.src[0];
  const DstDesc& d = a.dst[0];
  if (s.up || s.C != a.Cin || s.c0 != 0 || s.H != a.IH || s.W != a.IW || a.IH != a.OH || a.IW != a.OW) return false;
  if (d.pool || d.c0 != 0 || d.H != a.OH || d.W != a.OW || d.cn != a.Cout) return false;
  if (a.Cin % 64 != 0 || a.Cin > 4096 || a.Cout % 8 != 0 || a.Cout < 8) return false;
  if ((long long)a.N * a.OH * a.OW < 1) return false;
  return true;
}

int gemm1x1_rows(const ConvArgs& a) { return g1_geom(a).rows; }

template <typename T, int NT, bool AFF, bool BRES>
static hipError_t g1_launch_k(const G1Args& ga, const G1Geom& g, hipStream_t st) {
  static bool set = false;
  if (!set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm1x1_kernel<T, NT, AFF, BRES>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    set = true;
  }
  hipLaunchKernelGGL((gemm1x1_kernel<T, NT, AFF, BRES>), dim3(g.G), dim3(256), g.lds, st, ga);
  return hipGetLastError();
}
template <typename T, int NT>
static hipError_t g1_launch(const G1Args& ga, const G1Geom& g, bool aff, hipStream_t st) {
  if constexpr (NT == 4) {
    if (g.bres) return aff ? g1_launch_k<T, NT, true, true>(ga, g, st) : g1_launch_k<T, NT, false, true>(ga, g, st);
  }
  return aff ? g1_launch_k<T, NT, true, false>(ga, g, st) : g1_launch_k<T, NT, false, false>(ga, g, st);
}

hipError_t launch_gemm1x1(int dtype, const ConvArgs& a, hipStream_t st) {
  const G1Geom g = g1_geom(a);
  G1Args ga;
  const SrcDesc& s = a.src[0];
  const DstDesc& d = a.dst[0];
  ga.x = (const char*)s.ptr; ga.xstride = (long long)s.C * 2;
  ga.scale = s.scale; ga.shift = s.shift; ga.relu = s.relu;
  ga.w = (const char*)a.W;
  ga.y = (char*)d.ptr; ga.ystride = (long long)d.C * 2; ga.accum = (d.accum || a.out_mode == OUT_ACCUM) ? 1 : 0;
  ga.M = (int)((long long)a.N * a.OH * a.OW); ga.K = a.Cin; ga.N = a.Cout; ga.n_tiles = g.n_tiles; ga.m_tiles = g.m_tiles;
  ga.slab = a.stat_slab; ga.slab_row0 = a.slab_row0;
  ga.bias = a.bias; ga.relu_out = a.relu_out;
  const bool aff = s.scale != nullptr;
  if (dtype == DT_F16) {
    switch (g.NT) {
      case 4: return g1_launch<f16_t, 4>(ga, g, aff, st);
      case 2: return g1_launch<f16_t, 2>(ga, g, aff, st);
      default: return g1_launch<f16_t, 1>(ga, g, aff, st);
    }
  }
  switch (g.NT) {
    case 4: return g1_launch<bf16_t, 4>(ga, g, aff, st);
    case 2: return g1_launch<bf16_t, 2>(ga, g, aff, st);
    default: return g1_launch<bf16_t, 1>(ga, g, aff, st);
  }
}

}  // namespace octseg
