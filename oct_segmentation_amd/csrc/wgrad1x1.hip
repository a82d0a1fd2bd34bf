// wgrad1x1.hip -- weight gradients of stride-1 1x1 convolutions (ResNet bottleneck conv1 / conv3 / stride-1 shortcuts, LinkNet / FPN /
// ASPP pointwise convs) on gfx950, bf16:
//
//   dW[co][ci] = sum_p dy[p][co] * f(x[p][ci]),   p over the N*H*W pixels,   f = the consumer's lazy BatchNorm + ReLU (or identity)
//
// Round 3's counters (profiles/r4_sq_counters_wgrad1_*.txt) showed the generic kernel on these shapes at 10 % MFMA-busy and 1.2 TB/s: every
// bottleneck shape took 60-64 us whatever its size because a workgroup's tile loop was a chain of dependent round trips (load the next
// 128 pixels into registers, 8 MFMAs, barrier, store, barrier).  This kernel keeps the bytes in flight instead:
//   * both operands travel global -> LDS by LDS-DMA (global_load_lds_dwordx4, no registers, no VALU) into a ring of NST stages of 64
//     pixels, D = NST - 1 stages in flight per workgroup, two workgroups per CU: 96-128 KB of loads in flight per CU;
//   * the tiles lie pixel-major as in HBM (128-byte rows), 16-byte chunks XOR-swizzled by pixel-row bit 1 (applied on the GLOBAL side of
//     the DMA, whose LDS side is linear), so that the transposed fragment reads (ds_read_b64_tr_b16) are bank-conflict free at pitch 128;
//   * wave tile (32 MB) co x 32 ci, MB = 2 where Cout allows: the x fragment -- the one that may need the lazy affine -- feeds two MFMAs;
//   * the lazy BatchNorm + ReLU is applied to the x FRAGMENT in registers: after the transposed read a lane holds 8 pixels of ONE channel,
//     so scale / shift are two per-lane constants for the whole kernel (scalar v_fma_f32: a packed-f32 instruction in an MFMA gap costs
//     more than the two it replaces, MI355X_MICROARCH.md), rounded to bf16 and clamped exactly as the forward's staging does;
//   * one barrier per stage; s_waitcnt vmcnt counted by hand (the DMAs are inline asm, invisible to hipcc's wait model; nothing else in
//     the loop touches global memory);
//   * split-K over pixel ranges, a range kept on one XCD (its tiles share that L2), fp32 atomics shaped as 128-byte row segments.
// Reference: the weight gradients autograd computes for nn.Conv2d(k=1) inside smp's encoders / decoders under
// OCTSegmentationModel.training_step + loss.backward() (src/models/smp/model.py:73-95).
#include "common.h"
#include "conv_common.h"
#include "kernels.h"

#include <cstdlib>

namespace octseg {

typedef __attribute__((ext_vector_type(4))) short w1_s16x4_t;
typedef __attribute__((address_space(3))) w1_s16x4_t w1_lds_s16x4_t;

constexpr int W1_SP = 64;                 // pixels per stage
constexpr int W1_PLANE = W1_SP * 128;     // one operand plane of a stage: 64 pixels x 64 channels x 2 bytes

struct Wgrad1x1Args {
  const void* dy; int dyC;                // [P][dyC] bf16
  SrcDesc src[MAX_SRC]; int nsrc;         // x sources (flattened: H = rows of 16, W = 16; up = 0)
  long long P;                            // pixels (multiple of 16)
  int Cin, Cout;
  float* dW;                              // [Cout][Cin] fp32 (tap slab already applied), accumulated with atomics
  int spw;                                // stages per workgroup (split-K)
  int co_fast;                            // co tiles fastest in the XCD-local order (speed only)
};

// DEEP = 0: ring of 3 x 24 KB (MB = 2) / 4 x 16 KB (MB = 1), two workgroups per CU; DEEP = 1: 6 x 24 KB / 8 x 16 KB, one workgroup per CU
// (half the workgroups = half the split-K atomic traffic, which runs at 1.3 TB/s chip-wide, memory side)
template <int MB, bool DEEP> struct W1Ring { static constexpr int NST = (MB == 2 ? 3 : 4) * (DEEP ? 2 : 1); };

template <int MB, bool AFF, bool DEEP>
__global__ __launch_bounds__(256) void wgrad1x1_kernel(const Wgrad1x1Args a) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  constexpr int NST = W1Ring<MB, DEEP>::NST;
  constexpr int D = NST - 1;                      // stages in flight
  constexpr int NPL = MB + 1;                     // planes per stage: MB of dy, one of x
  constexpr int STAGE = NPL * W1_PLANE;
  constexpr int IPW = NPL * 2;                    // DMA instructions per wave and stage (a plane = 8 wave-instructions of 1 KB)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wq_m = wave >> 1, wq_n = wave & 1;

  // XCD-aware placement: all (ci, co) tiles of one pixel range read the same dy / x rows -- keep a range on ONE XCD (one L2)
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
  const int ci0 = bx * 64, co0 = by * (64 * MB);
  const long long ns_total = (a.P + W1_SP - 1) / W1_SP;
  const long long s_begin = (long long)bz * a.spw;
  const long long s_end = min(s_begin + a.spw, ns_total);
  const int ns = (int)(s_end - s_begin);

  // x source of this ci tile (sources start on 64-channel boundaries: host-checked)
  const SrcDesc* sx = &a.src[0];
#pragma unroll
  for (int i = 1; i < MAX_SRC; ++i) if (i < a.nsrc && ci0 >= a.src[i].c0) sx = &a.src[i];
  const int xC = sx->C, xcl = ci0 - sx->c0;

  // ---- DMA addressing: lane l of a wave-instruction fills LDS bytes [16 l, 16 l + 16) of a 1 KB piece = pixel row rr = l >> 3 of the
  // piece, physical chunk l & 7, which holds LOGICAL chunk (l & 7) ^ swz(rr)
  const int rr = lane >> 3;
  const int lchunk = (lane & 7) ^ (((rr >> 1) & 1) << 2);
  // piece j (0 .. 2 NPL - 1) of this wave: plane = j >> 1, rows (wave * 2 + (j & 1)) * 8 .. + 8
  const char* gp[IPW];                 // per-lane global address of piece j at stage 0 of this workgroup
  unsigned row_bytes[IPW];
  long long prow[IPW];                 // pixel row of piece j at stage s_begin
#pragma unroll
  for (int j = 0; j < IPW; ++j) {
    const int plane = j >> 1;
    const int r = (wave * 2 + (j & 1)) * 8 + rr;
    prow[j] = s_begin * W1_SP + r;
    if (plane < MB) {
      row_bytes[j] = (unsigned)a.dyC * 2u;
      gp[j] = (const char*)a.dy + ((size_t)(co0 + plane * 64 + lchunk * 8)) * 2;
    } else {
      row_bytes[j] = (unsigned)xC * 2u;
      gp[j] = (const char*)sx->ptr + ((size_t)(xcl + lchunk * 8)) * 2;
    }
  }
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const long long last_row = a.P - 1;
  auto issue = [&](int s) {            // stage s (relative) into ring slot s % NST
    const unsigned slot = lds_base + (unsigned)(s % NST) * STAGE;
#pragma unroll
    for (int j = 0; j < IPW; ++j) {
      long long pr = prow[j] + (long long)s * W1_SP;
      pr = pr < last_row ? pr : last_row;           // rows past the end: clamped (their k-steps are never contracted)
      const char* g = gp[j] + (size_t)pr * row_bytes[j];
      const unsigned dst = __builtin_amdgcn_readfirstlane(slot + (unsigned)(j >> 1) * W1_PLANE + (unsigned)(wave * 2 + (j & 1)) * 1024u);
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
    }
  };

  // ---- fragment addressing (ds_read_b64_tr_b16): 16 lanes read a 4-pixel x 16-channel block, lane i = (q, p) supplies pixel row q,
  // channels 4p .. 4p + 3; afterwards lane i holds channel i of the block for the 4 pixels.  hh = k half (pixels 8 hh ..), cb = channel
  // half of the 32-channel MFMA operand.
  const int g4 = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3, hh = g4 >> 1, cb = g4 & 1;
  const int swz = ((q >> 1) & 1) << 2;
  auto frag_off = [&](int chunk_base) {           // byte offset inside a plane, k-step 0, first 4 pixels
    const int cL = chunk_base + cb * 2 + (p >> 1);
    return (8 * hh + q) * 128 + ((cL ^ swz) << 4) + (p & 1) * 8;
  };
  int yoff[MB];
#pragma unroll
  for (int m = 0; m < MB; ++m) yoff[m] = (MB == 2 ? wq_m * W1_PLANE : 0) + frag_off(MB == 2 ? m * 4 : wq_m * 4);   // MB = 2: dy plane wq_m, 32-channel block m
  const int xoff = MB * W1_PLANE + frag_off(wq_n * 4);

  // lazy BatchNorm of the x operand: this lane's channel is fixed for the whole kernel
  float sc = 1.f, sh = -0.f;
  unsigned fl16 = 0x80008000u;
  if constexpr (AFF) {
    if (sx->scale != nullptr) {
      const int c = xcl + wq_n * 32 + (lane & 31);
      sc = sx->scale[c]; sh = sx->shift[c];
      if (sx->relu) fl16 = 0u;
    }
  }

  f32x16_t acc[MB];
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[m][k] = 0.f;

  struct Pair { w1_s16x4_t lo, hi; };
  auto read_frag = [&](const char* base) -> uint4 {
    Pair f;
    f.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((w1_lds_s16x4_t*)(base));
    f.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((w1_lds_s16x4_t*)(base + 4 * 128));
    return __builtin_bit_cast(uint4, f);
  };
  auto affine_frag = [&](uint4 v) -> uint4 {
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    typedef __attribute__((ext_vector_type(2))) short s16x2_t;
    unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x2_t x = {__uint_as_float(w[i] << 16), __uint_as_float(w[i] & 0xffff0000u)};
      asm("v_fma_f32 %0, %1, %2, %3" : "=v"(x[0]) : "v"(x[0]), "v"(sc), "v"(sh));
      asm("v_fma_f32 %0, %1, %2, %3" : "=v"(x[1]) : "v"(x[1]), "v"(sc), "v"(sh));
      const bf16x2_t b = __builtin_convertvector(x, bf16x2_t);
      const s16x2_t mx = __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, b), __builtin_bit_cast(s16x2_t, fl16));
      w[i] = __builtin_bit_cast(unsigned, mx);
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
  };

  auto load_k = [&](const char* st, int kk, uint4& xf, uint4 (&yf)[MB]) {
    xf = read_frag(st + xoff + kk * 2048);
#pragma unroll
    for (int m = 0; m < MB; ++m) yf[m] = read_frag(st + yoff[m] + kk * 2048);
  };
  auto contract = [&](uint4 xf, const uint4 (&yf)[MB]) {
    if constexpr (AFF) xf = affine_frag(xf);
#pragma unroll
    for (int m = 0; m < MB; ++m) Tr<bf16_t>::mma(yf[m], xf, acc[m]);
  };

  if (ns > 0) {
    // scale / shift are plain loads: retire them HERE (hipcc would otherwise put its s_waitcnt vmcnt(0) at their first use, inside
    // the loop, every iteration -- draining the DMA ring it knows nothing about)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(sc), "+v"(sh));
#pragma unroll
    for (int s = 0; s < D; ++s) if (s < ns) issue(s);
    for (int s = 0; s < ns; ++s) {
      // this wave's DMAs of stage s have landed once at most the younger stages' are outstanding
      const int younger = min(D - 1, ns - 1 - s);
      switch (younger) {                   // (immediates: one s_waitcnt per case)
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(1 * IPW) : "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * IPW) : "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * IPW) : "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * IPW) : "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(5 * IPW) : "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(6 * IPW) : "memory"); break;
      }
      __syncthreads();                     // every wave's pieces of stage s are in LDS; everybody is done with stage s - 1
      if (s + D < ns) issue(s + D);        // into the slot stage s - 1 used
      const char* st = smem + (s % NST) * STAGE;
      const long long pix0 = (s_begin + s) * W1_SP;
      const int nk = (int)min((long long)4, (a.P - pix0) >> 4);
      if (nk == 4) {        // whole stage: fragments of k-step kk + 1 requested before the MFMAs of kk
        uint4 xa, xb, ya[MB], yb[MB];
        load_k(st, 0, xa, ya);
        load_k(st, 1, xb, yb);
        contract(xa, ya);
        load_k(st, 2, xa, ya);
        contract(xb, yb);
        load_k(st, 3, xb, yb);
        contract(xa, ya);
        contract(xb, yb);
      } else {              // the last stage of the pixel list may hold fewer than four 16-pixel k-steps
        for (int kk = 0; kk < nk; ++kk) {
          uint4 xa, ya[MB];
          load_k(st, kk, xa, ya);
          contract(xa, ya);
        }
      }
    }
  }

  // ---- combine: fp32 atomics, lanes 0-31 cover 128 contiguous bytes of one dW row ----
  const int ci = ci0 + wq_n * 32 + (lane & 31);
#pragma unroll
  for (int m = 0; m < MB; ++m) {
    const int cob = co0 + (MB == 2 ? wq_m * 64 + m * 32 : wq_m * 32);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int co = cob + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
      atomicAdd(a.dW + (size_t)co * a.Cin + ci, acc[m][i]);
    }
  }
}

static bool wgrad1x1_enabled() {
  static const bool off = getenv("OCTSEG_NO_WGRAD1X1") != nullptr;   // A/B switch
  return !off;
}

// `a` must already be flattened by flatten_1x1 (wgrad_mfma.hip): N = 1, OW = 16 = every W, OH = rows
bool wgrad1x1_eligible(const WgradArgs& a, int dtype) {
  if (!wgrad1x1_enabled() || dtype != DT_BF16) return false;
  if (a.ntaps != 1 || a.istride != 1 || a.dstride != 1 || a.tap_dy[0] != 0 || a.tap_dx[0] != 0 || a.doy != 0 || a.dox != 0) return false;
  if (a.N != 1 || a.OW != TW || a.IW != TW || a.DW != TW || a.IH != a.OH || a.DH != a.OH) return false;
  if (a.Cin % 64 != 0 || a.Cout % 64 != 0 || a.dyC % 8 != 0 || a.Cout > a.dyC) return false;
  if ((long long)a.OH * TW < 1024) return false;                     // tiny maps: the pipeline never fills
  for (int i = 0; i < a.nsrc; ++i) {
    const SrcDesc& s = a.src[i];
    if (s.up || s.H != a.IH || s.W != a.IW || s.c0 % 64 != 0 || s.C % 8 != 0) return false;
    const int cn = (i + 1 < a.nsrc ? a.src[i + 1].c0 : a.Cin) - s.c0;   // channels of the concatenation this source supplies
    if (cn % 64 != 0 || cn > s.C) return false;
    if (((size_t)s.ptr & 15) != 0) return false;
  }
  if (((size_t)a.dy & 15) != 0) return false;
  return true;
}

template <int MB, bool AFF, bool DEEP>
static hipError_t launch_w1(const Wgrad1x1Args& a, int gx, int gy, int gz, hipStream_t st) {
  constexpr int NST = W1Ring<MB, DEEP>::NST;
  constexpr size_t lds = (size_t)NST * (MB + 1) * W1_PLANE;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)wgrad1x1_kernel<MB, AFF, DEEP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL((wgrad1x1_kernel<MB, AFF, DEEP>), dim3(gx, gy, gz), dim3(256), lds, st, a);
  return hipGetLastError();
}

hipError_t launch_wgrad1x1(int dtype, const WgradArgs& w, hipStream_t st) {
  (void)dtype;
  Wgrad1x1Args a;
  a.dy = w.dy; a.dyC = w.dyC; a.nsrc = w.nsrc; a.Cin = w.Cin; a.Cout = w.Cout;
  for (int i = 0; i < MAX_SRC; ++i) a.src[i] = w.src[i];
  a.P = (long long)w.OH * TW;
  a.dW = w.dW + (size_t)w.tap_w[0] * w.Cout * w.Cin;
  bool aff = false;
  for (int i = 0; i < w.nsrc; ++i) aff = aff || w.src[i].scale != nullptr;
  const int MB = w.Cout % 128 == 0 ? 2 : 1;
  const int gx = w.Cin / 64, gy = w.Cout / (64 * MB);
  const long long ns_total = (a.P + W1_SP - 1) / W1_SP;
  // split-K: two workgroups per CU in one resident round; every workgroup ends by adding its tile to dW with atomics (ksplit x the
  // size of dW in atomic traffic), so no more ranges than that -- and at least 4 stages per range, or the ring never fills
  static const int wg_env = getenv("OCTSEG_WGRAD1X1_WGS") ? atoi(getenv("OCTSEG_WGRAD1X1_WGS")) : 0;
  static const int deep_env = getenv("OCTSEG_WGRAD1X1_DEEP") ? atoi(getenv("OCTSEG_WGRAD1X1_DEEP")) : -1;
  const bool deep = deep_env >= 0 ? deep_env != 0 : false;
  const int wg_target = wg_env > 0 ? wg_env : (deep ? 256 : 512);
  long long ks = wg_target / (gx * gy);
  if (ks > ns_total / 4) ks = ns_total / 4;
  if (ks < 1) ks = 1;
  if (deterministic_mode()) ks = 1;      // one writer per dW element: a fixed summation order
  const long long spw = (ns_total + ks - 1) / ks;
  ks = (ns_total + spw - 1) / spw;
  a.spw = (int)spw;
  static const bool ci_major = getenv("OCTSEG_WGRAD_CI_MAJOR") != nullptr;   // A/B switch
  a.co_fast = ci_major ? 0 : 1;
  if (deep) {
    if (MB == 2) return aff ? launch_w1<2, true, true>(a, gx, gy, (int)ks, st) : launch_w1<2, false, true>(a, gx, gy, (int)ks, st);
    return aff ? launch_w1<1, true, true>(a, gx, gy, (int)ks, st) : launch_w1<1, false, true>(a, gx, gy, (int)ks, st);
  }
  if (MB == 2) return aff ? launch_w1<2, true, false>(a, gx, gy, (int)ks, st) : launch_w1<2, false, false>(a, gx, gy, (int)ks, st);
  return aff ? launch_w1<1, true, false>(a, gx, gy, (int)ks, st) : launch_w1<1, false, false>(a, gx, gy, (int)ks, st);
}

}  // namespace octseg
