// conv_common.h -- device helpers shared by the conv and wgrad MFMA kernels.
#pragma once
#include "common.h"

namespace octseg {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

constexpr int TW = 16;                // output-grid tile width (pixels)
constexpr int NTHR = 256;

// max(v, lo) that keeps a NaN a NaN (fmaxf / v_max_f32 return the OTHER operand, so a diverged accumulator would come out as 0 -- or as
// -FLT_MAX without a ReLU -- where torch's relu / plain store propagate it) and, with lo = -inf for "no ReLU", leaves -inf alone
static __device__ __forceinline__ float clamp_lo(float v, float lo) { return v < lo ? lo : v; }
constexpr float NO_FLOOR = -__builtin_inff();

template <typename T> struct Tr;
template <> struct Tr<float> {
  static constexpr int VEC = 4;  // channels per 16-byte vector
  static __device__ __forceinline__ void mma(const uint4& a, const uint4& b, f32x16_t& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
  }
  // relu?(x*scale+shift) on a 4-channel vector
  static __device__ __forceinline__ uint4 affine(uint4 v, const float* sc, const float* sh, int relu) {
    float x[4] = {__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      x[i] = fmaf(x[i], sc[i], sh[i]);
      if (relu) x[i] = clamp_lo(x[i], 0.f);
    }
    return make_uint4(__float_as_uint(x[0]), __float_as_uint(x[1]), __float_as_uint(x[2]), __float_as_uint(x[3]));
  }
  // branch-free form: max(x*scale+shift, lo), lo = 0 (ReLU) or anything below zero (none: taken as -inf)
  static __device__ __forceinline__ uint4 affine_lo(uint4 v, const float* sc, const float* sh, float lo) {
    float x[4] = {__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = clamp_lo(fmaf(x[i], sc[i], sh[i]), lo);
    return make_uint4(__float_as_uint(x[0]), __float_as_uint(x[1]), __float_as_uint(x[2]), __float_as_uint(x[3]));
  }
  static __device__ __forceinline__ float load(const void* p, size_t i) { return ((const float*)p)[i]; }
  static __device__ __forceinline__ void store(void* p, size_t i, float v) { ((float*)p)[i] = v; }
  // 16-bit helpers of the 2-byte specialisations: present so that `if (sizeof(T) == 4) ... else ...` bodies compile; never executed
  static __device__ __forceinline__ float lo(unsigned) { return 0.f; }
  static __device__ __forceinline__ float hi(unsigned) { return 0.f; }
  static __device__ __forceinline__ unsigned pk(float, float) { return 0u; }
  static __device__ __forceinline__ unsigned short bits16(float) { return 0; }
};
static __device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
  // one v_cvt_pk_bf16_f32 (two scalar conversions + shift + or took four instructions)
  typedef __attribute__((ext_vector_type(2))) float f32x2_t;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  const f32x2_t x = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2_t));
}
template <> struct Tr<bf16_t> {
  static constexpr int VEC = 8;
  static __device__ __forceinline__ void mma(const uint4& a, const uint4& b, f32x16_t& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
  // 16 x 16 x 32: lane (r = lane & 15, g = lane >> 4) holds row r, k = 8g .. 8g + 7 of both operands; C: column lane & 15, row 4g + reg.
  // Same cycles per FLOP as 32x32x16, but the shape holds a higher clock under the package power limit (MI355X_MICROARCH.md, DVFS (7))
  static __device__ __forceinline__ void mma16(const uint4& a, const uint4& b, f32x4_t& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
  // relu?(x*scale+shift) on 8 bf16 channels, 5 vector instructions per channel pair: two unpacks, v_pk_fma_f32,
  // v_cvt_pk_bf16_f32 and the ReLU as v_pk_max_i16 on the ROUNDED pair (a negative bf16 is a negative int16 and
  // rounding keeps the sign, so max(round(x), 0) == round(max(x, 0)) bit for bit; floor -32768 = no ReLU).
  // (The scalar form -- fma, max, select, convert, shift, or per channel -- cost 2.5x the instructions, and the
  // 1x1 loops were bound by exactly these: 400 vector instructions per 16 MFMAs.)
  static __device__ __forceinline__ unsigned affine_floor1(unsigned w, float s0, float s1, float t0, float t1, unsigned floor16) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    typedef __attribute__((ext_vector_type(2))) short s16x2_t;
    f32x2_t x = {__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
#ifdef OCTSEG_SCALAR_AFFINE
    // Two v_fma_f32 instead of one v_pk_fma_f32: beside MFMAs a packed-f32 VALU instruction costs ~22 cycles more than the two scalar
    // ones it replaces (MI355X_MICROARCH.md, 'price of one filler beside MFMAs').  asm so that -O3's SLP pass cannot re-pack them.
    // A translation unit whose staging runs in MFMA gaps defines OCTSEG_SCALAR_AFFINE before including this header.
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(x[0]) : "v"(x[0]), "v"(s0), "v"(t0));
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(x[1]) : "v"(x[1]), "v"(s1), "v"(t1));
#else
    const f32x2_t s2 = {s0, s1}, t2 = {t0, t1};
    x = __builtin_elementwise_fma(x, s2, t2);
#endif
    const bf16x2_t b = __builtin_convertvector(x, bf16x2_t);
    const s16x2_t m = __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, b), __builtin_bit_cast(s16x2_t, floor16));
    return __builtin_bit_cast(unsigned, m);
  }
  static __device__ __forceinline__ uint4 affine_floor(uint4 v, const float* sc, const float* sh, unsigned floor16) {
    unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = affine_floor1(w[i], sc[2 * i], sc[2 * i + 1], sh[2 * i], sh[2 * i + 1], floor16);
    return make_uint4(w[0], w[1], w[2], w[3]);
  }
  static __device__ __forceinline__ uint4 affine(uint4 v, const float* sc, const float* sh, int relu) {
    return affine_floor(v, sc, sh, relu ? 0u : 0x80008000u);
  }
  // lo = 0 (ReLU) or -FLT_MAX (none), as the fp32 specialisation takes it
  static __device__ __forceinline__ uint4 affine_lo(uint4 v, const float* sc, const float* sh, float lo) {
    return affine_floor(v, sc, sh, lo == 0.f ? 0u : 0x80008000u);
  }
  // the two halves of a packed pair as f32, a pair from two f32 (round to nearest even), one value's 16 bits
  static __device__ __forceinline__ float lo(unsigned w) { return __uint_as_float(w << 16); }
  static __device__ __forceinline__ float hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
  static __device__ __forceinline__ unsigned pk(float l, float h) { return pack_bf16(l, h); }
  static __device__ __forceinline__ unsigned short bits16(float v) { __bf16 b = (__bf16)v; return __builtin_bit_cast(unsigned short, b); }
  static __device__ __forceinline__ void unpack8(const uint4& v, float* x) {
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { x[2 * i] = __uint_as_float(w[i] << 16); x[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
  }
  static __device__ __forceinline__ uint4 pack8(const float* x) {
    return make_uint4(pack_bf16(x[0], x[1]), pack_bf16(x[2], x[3]), pack_bf16(x[4], x[5]), pack_bf16(x[6], x[7]));
  }
  static __device__ __forceinline__ float load(const void* p, size_t i) {
    return __uint_as_float((unsigned)((const bf16_t*)p)[i] << 16);
  }
  static __device__ __forceinline__ void store(void* p, size_t i, float v) {
    __bf16 b = (__bf16)v;
    ((bf16_t*)p)[i] = __builtin_bit_cast(unsigned short, b);
  }
};

// IEEE half storage (serving, BASELINE config #5): same kernels, v_mfma_f32_32x32x16_f16 (the bf16 form's rate), f32 accumulate.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
template <> struct Tr<f16_t> {
  static constexpr int VEC = 8;
  static __device__ __forceinline__ void mma(const uint4& a, const uint4& b, f32x16_t& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ void mma16(const uint4& a, const uint4& b, f32x4_t& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ float lo(unsigned w) { return (float)__builtin_bit_cast(f16x2_t, w)[0]; }
  static __device__ __forceinline__ float hi(unsigned w) { return (float)__builtin_bit_cast(f16x2_t, w)[1]; }
  static __device__ __forceinline__ unsigned pk(float l, float h) {
    const f16x2_t v = {(_Float16)l, (_Float16)h};   // v_cvt_f16_f32 rounds to nearest even (v_cvt_pkrtz would truncate)
    return __builtin_bit_cast(unsigned, v);
  }
  static __device__ __forceinline__ unsigned short bits16(float v) { const _Float16 x = (_Float16)v; return __builtin_bit_cast(unsigned short, x); }
  // relu?(x * scale + shift) on 8 channels; floor16 = 0: ReLU (v_pk_max_f16 on the rounded pair), anything else: none
  static __device__ __forceinline__ unsigned affine_floor1(unsigned w, float s0, float s1, float t0, float t1, unsigned floor16) {
    const f16x2_t zero = {(_Float16)0.f, (_Float16)0.f};
    const float l = fmaf(lo(w), s0, t0), h = fmaf(hi(w), s1, t1);
    f16x2_t p = {(_Float16)l, (_Float16)h};
    if (floor16 == 0u) p = __builtin_elementwise_max(p, zero);
    return __builtin_bit_cast(unsigned, p);
  }
  static __device__ __forceinline__ uint4 affine_floor(uint4 v, const float* sc, const float* sh, unsigned floor16) {
    unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = affine_floor1(w[i], sc[2 * i], sc[2 * i + 1], sh[2 * i], sh[2 * i + 1], floor16);
    return make_uint4(w[0], w[1], w[2], w[3]);
  }
  static __device__ __forceinline__ uint4 affine(uint4 v, const float* sc, const float* sh, int relu) {
    return affine_floor(v, sc, sh, relu ? 0u : 0x80008000u);
  }
  static __device__ __forceinline__ uint4 affine_lo(uint4 v, const float* sc, const float* sh, float lo_) {
    return affine_floor(v, sc, sh, lo_ == 0.f ? 0u : 0x80008000u);
  }
  static __device__ __forceinline__ void unpack8(const uint4& v, float* x) {
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { x[2 * i] = lo(w[i]); x[2 * i + 1] = hi(w[i]); }
  }
  static __device__ __forceinline__ uint4 pack8(const float* x) { return make_uint4(pk(x[0], x[1]), pk(x[2], x[3]), pk(x[4], x[5]), pk(x[6], x[7])); }
  static __device__ __forceinline__ float load(const void* p, size_t i) {
    return (float)__builtin_bit_cast(_Float16, ((const unsigned short*)p)[i]);
  }
  static __device__ __forceinline__ void store(void* p, size_t i, float v) { ((unsigned short*)p)[i] = bits16(v); }
};

// Per-thread view of the source its channel vector falls into (resolved once per chunk).
struct SrcSel {
  const char* ptr; const float* scale; const float* shift;
  int C, cl, H, W, up, relu;
};
static __device__ __forceinline__ SrcSel select_src(const SrcDesc* src, int nsrc, int c) {
  SrcSel s;
  s.ptr = (const char*)src[0].ptr; s.scale = src[0].scale; s.shift = src[0].shift;
  s.C = src[0].C; s.cl = c - src[0].c0; s.H = src[0].H; s.W = src[0].W; s.up = src[0].up; s.relu = src[0].relu;
#pragma unroll
  for (int i = 1; i < MAX_SRC; ++i) {
    if (i < nsrc && c >= src[i].c0) {
      s.ptr = (const char*)src[i].ptr; s.scale = src[i].scale; s.shift = src[i].shift;
      s.C = src[i].C; s.cl = c - src[i].c0; s.H = src[i].H; s.W = src[i].W; s.up = src[i].up; s.relu = src[i].relu;
    }
  }
  return s;
}

// Stage the input window of one channel chunk into LDS (gather + lazy BN/ReLU).
// Window pixel hp=(hy,hx) <-> virtual input (gy0 + hy*smul, gx0 + hx*smul).
template <typename T, int RB>
static __device__ __forceinline__ void stage_window(char* lds, const SrcDesc* src, int nsrc, int Cin,
                                                    int chunk, int n, int gy0, int gx0, int smul, int RW,
                                                    int npix, float inv_rw, int IH, int IW, int tid) {
  constexpr int VEC = Tr<T>::VEC;
  constexpr int KC = RB / (int)sizeof(T);   // channels per LDS row
  constexpr int VPR = RB / 16;              // 16-byte vectors per LDS row
  constexpr int PSTEP = NTHR / VPR;         // window pixels covered per pass
  constexpr int PITCH = RB + 16;            // padded row pitch (bank spread for wide LDS reads)
  const int cv = tid % VPR;
  const int c = chunk * KC + cv * VEC;
  const bool cvalid = c < Cin;
  SrcSel s = select_src(src, nsrc, cvalid ? c : 0);
  float sc[VEC], sh[VEC];
  const bool has_aff = cvalid && s.scale != nullptr;
  if (has_aff) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) { sc[i] = s.scale[s.cl + i]; sh[i] = s.shift[s.cl + i]; }
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) { sc[i] = 1.f; sh[i] = 0.f; }
  }
  constexpr int U = 4;
  for (int base = tid / VPR; base < npix; base += PSTEP * U) {
    uint4 v[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int hp = base + PSTEP * u;
      const int hy = (int)(((float)hp + 0.5f) * inv_rw);
      const int hx = hp - hy * RW;
      const int iy = gy0 + hy * smul, ix = gx0 + hx * smul;
      ok[u] = cvalid && hp < npix && (unsigned)iy < (unsigned)IH && (unsigned)ix < (unsigned)IW;
      v[u] = make_uint4(0, 0, 0, 0);
      if (ok[u]) {
        const size_t e = (((size_t)n * s.H + (iy >> s.up)) * s.W + (ix >> s.up)) * s.C + s.cl;
        v[u] = *(const uint4*)(s.ptr + e * sizeof(T));
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int hp = base + PSTEP * u;
      if (hp < npix) {
        uint4 w = v[u];
        if (ok[u] && has_aff) w = Tr<T>::affine(w, sc, sh, s.relu);
        *(uint4*)(lds + hp * PITCH + cv * 16) = w;
      }
    }
  }
}


// Pass-wise stager of the input window: pass p covers window pixels [p*PSTEP, (p+1)*PSTEP), one
// 16-byte channel vector per thread.  load() only issues the global load, write() applies the lazy
// BN/ReLU and stores to LDS -- the caller puts MFMAs in between so the HBM/L2 latency is hidden.
template <typename T, int RB, int NT_, int PITCH_ = RB + 16>
struct WindowStager {
  static constexpr int VEC = Tr<T>::VEC;
  static constexpr int KC = RB / (int)sizeof(T);
  static constexpr int VPR = RB / 16;
  static constexpr int PSTEP = NT_ / VPR;
  static constexpr int PITCH = PITCH_;
  SrcSel s;
  float sc[VEC], sh[VEC];
  bool cvalid, has_aff;
  unsigned fl16;          // packed 16-bit floor of write_at_nb: 0 = ReLU, 0x80008000 = none
  int cv, p0;
  const char* img_base;   // source pointer of (image n, this thread's channel vector): set by bind_image()
  int row_bytes, pix_bytes;
  __device__ __forceinline__ void bind_image(int n) {
    img_base = s.ptr + ((size_t)n * s.H * s.W * s.C + s.cl) * sizeof(T);
    pix_bytes = s.C * (int)sizeof(T);
    row_bytes = s.W * pix_bytes;
  }
  // load of window pixel (hy, hx) (already decomposed by the caller: no division in the pipeline);
  // 32-bit offset inside the image, clamped address, validity resolved at write time
  __device__ __forceinline__ uint4 load_at(int hy, int hx, bool in_window, int gy0, int gx0, int smul, int IH, int IW,
                                           bool& ok) const {
    const int iy = gy0 + hy * smul, ix = gx0 + hx * smul;
    ok = cvalid && in_window && (unsigned)iy < (unsigned)IH && (unsigned)ix < (unsigned)IW;
    const int iyc = min(max(iy, 0), IH - 1), ixc = min(max(ix, 0), IW - 1);
    const unsigned off = (unsigned)((iyc >> s.up) * row_bytes + (ixc >> s.up) * pix_bytes);
    return *(const uint4*)(img_base + off);
  }
  // load_at with the window pixel already resolved to clamped image coordinates (they do not depend on the K chunk)
  __device__ __forceinline__ uint4 load_xy(int iyc, int ixc) const {
    const unsigned off = (unsigned)((iyc >> s.up) * row_bytes + (ixc >> s.up) * pix_bytes);
    return *(const uint4*)(img_base + off);
  }
  __device__ __forceinline__ const char* addr_xy(int iyc, int ixc) const {
    return img_base + (unsigned)((iyc >> s.up) * row_bytes + (ixc >> s.up) * pix_bytes);
  }
  // address half of load_at (for callers that issue the load themselves)
  __device__ __forceinline__ const char* addr_at(int hy, int hx, bool in_window, int gy0, int gx0, int smul, int IH, int IW,
                                                 bool& ok) const {
    const int iy = gy0 + hy * smul, ix = gx0 + hx * smul;
    ok = cvalid && in_window && (unsigned)iy < (unsigned)IH && (unsigned)ix < (unsigned)IW;
    const int iyc = min(max(iy, 0), IH - 1), ixc = min(max(ix, 0), IW - 1);
    const unsigned off = (unsigned)((iyc >> s.up) * row_bytes + (ixc >> s.up) * pix_bytes);
    return img_base + off;
  }
  // value half of write_at: lazy BN / ReLU, zero outside the image
  __device__ __forceinline__ uint4 prep(uint4 v, bool ok) const {
    if (has_aff) v = Tr<T>::affine(v, sc, sh, s.relu);
    if (!ok) v = make_uint4(0, 0, 0, 0);
    return v;
  }
  __device__ __forceinline__ void write_at(char* lds_row, uint4 v, bool ok) const {
    if (has_aff) v = Tr<T>::affine(v, sc, sh, s.relu);
    if (!ok) v = make_uint4(0, 0, 0, 0);
    *(uint4*)(lds_row + cv * 16) = v;
  }
  // write_at without the branch on has_aff (16-bit types): the identity transform is applied instead, so that the whole tap of
  // conv3x3p is one basic block the scheduler can interleave with the MFMAs
  __device__ __forceinline__ void write_at_nb(char* lds_row, uint4 v, bool ok) const {
    v = Tr<T>::affine_floor(v, sc, sh, fl16);
    if (!ok) v = make_uint4(0, 0, 0, 0);
    *(uint4*)(lds_row + cv * 16) = v;
  }
  // uniform (host: every source starts on a K-chunk boundary): the chunk lies in ONE source, which is then picked on
  // the scalar unit; otherwise every lane selects for its own channel (4 x 12 vector selects per call)
  // select(): source, channel vector and validity of this thread for `chunk` -- no memory access
  __device__ __forceinline__ void select(const SrcDesc* src, int nsrc, int Cin, int chunk, int tid, bool uniform = false) {
    cv = tid % VPR;
    p0 = tid / VPR;
    const int c = chunk * KC + cv * VEC;
    cvalid = c < Cin;
    if (uniform) {
      s = select_src(src, nsrc, chunk * KC);
      s.cl = cvalid ? s.cl + cv * VEC : 0;
    } else {
      s = select_src(src, nsrc, cvalid ? c : 0);
    }
    has_aff = cvalid && s.scale != nullptr;
    fl16 = (has_aff && s.relu) ? 0u : 0x80008000u;
  }
  // setup(): select() + the lazy-BN parameters of the channel vector (plain loads: hipcc waits for them at first use)
  __device__ __forceinline__ void setup(const SrcDesc* src, int nsrc, int Cin, int chunk, int tid, bool uniform = false) {
    select(src, nsrc, Cin, chunk, tid, uniform);
    if (has_aff) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) { sc[i] = s.scale[s.cl + i]; sh[i] = s.shift[s.cl + i]; }
    } else {
#pragma unroll
      for (int i = 0; i < VEC; ++i) { sc[i] = 1.f; sh[i] = -0.f; }   // (x * 1 + -0 == x bit for bit, -0 included: write_at_nb)
    }
  }
  // Branch-free: the address is clamped into the tensor so the load is unconditional (the compiler can
  // then keep it in flight across the MFMA block); out-of-range elements are zeroed in write().
  __device__ __forceinline__ uint4 load(int pass, int n, int gy0, int gx0, int smul, int RW, int npix, float inv_rw,
                                        int IH, int IW, bool& ok) const {
    const int hp = pass * PSTEP + p0;
    const int hy = (int)(((float)hp + 0.5f) * inv_rw);
    const int hx = hp - hy * RW;
    const int iy = gy0 + hy * smul, ix = gx0 + hx * smul;
    ok = cvalid && hp < npix && (unsigned)iy < (unsigned)IH && (unsigned)ix < (unsigned)IW;
    const int iyc = min(max(iy, 0), IH - 1), ixc = min(max(ix, 0), IW - 1);
    const size_t e = (((size_t)n * s.H + (iyc >> s.up)) * s.W + (ixc >> s.up)) * s.C + s.cl;
    return *(const uint4*)(s.ptr + e * sizeof(T));
  }
  // unconditional store: the LDS window is padded to whole passes
  __device__ __forceinline__ void write(char* lds, int pass, uint4 v, bool ok) const {
    const int hp = pass * PSTEP + p0;
    if (has_aff) v = Tr<T>::affine(v, sc, sh, s.relu);
    if (!ok) v = make_uint4(0, 0, 0, 0);
    *(uint4*)(lds + hp * PITCH + cv * 16) = v;
  }
};

}  // namespace octseg
