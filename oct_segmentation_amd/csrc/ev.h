// ev.h -- 16-byte element-vector helpers shared by the HBM-bound kernels (elementwise.hip, fpn.hip): NHWC tensors are moved as
// 8 x bf16 / f16 or 4 x f32 per lane and processed in f32.
#pragma once
#include "common.h"

namespace octseg {

template <typename T> struct EV;  // 16-byte element vector helpers
template <> struct EV<float> {
  static constexpr int VEC = 4;
  static __device__ __forceinline__ void unpack(const uint4& v, float* x) {
    x[0] = __uint_as_float(v.x); x[1] = __uint_as_float(v.y); x[2] = __uint_as_float(v.z); x[3] = __uint_as_float(v.w);
  }
  static __device__ __forceinline__ uint4 pack(const float* x) {
    return make_uint4(__float_as_uint(x[0]), __float_as_uint(x[1]), __float_as_uint(x[2]), __float_as_uint(x[3]));
  }
};
static __device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {
  // one v_cvt_pk_bf16_f32 (two scalar conversions + shift + or took four instructions)
  typedef __attribute__((ext_vector_type(2))) float f32x2_t;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  const f32x2_t x = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2_t));
}
template <> struct EV<bf16_t> {
  static constexpr int VEC = 8;
  static __device__ __forceinline__ void unpack(const uint4& v, float* x) {
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { x[2 * i] = __uint_as_float(w[i] << 16); x[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
  }
  static __device__ __forceinline__ uint4 pack(const float* x) {
    return make_uint4(pk_bf16(x[0], x[1]), pk_bf16(x[2], x[3]), pk_bf16(x[4], x[5]), pk_bf16(x[6], x[7]));
  }
};
template <> struct EV<f16_t> {   // IEEE half (serving dtype): conversions round to nearest even
  static constexpr int VEC = 8;
  typedef __attribute__((ext_vector_type(2))) _Float16 h2_t;
  static __device__ __forceinline__ void unpack(const uint4& v, float* x) {
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { const h2_t p = __builtin_bit_cast(h2_t, w[i]); x[2 * i] = (float)p[0]; x[2 * i + 1] = (float)p[1]; }
  }
  static __device__ __forceinline__ uint4 pack(const float* x) {
    unsigned w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const h2_t p = {(_Float16)x[2 * i], (_Float16)x[2 * i + 1]}; w[i] = __builtin_bit_cast(unsigned, p); }
    return make_uint4(w[0], w[1], w[2], w[3]);
  }
};
// f16 is the dtype of eval forwards only: the training-only sweeps have no f16 instantiation and refuse it
#define OCTSEG_NO_F16(dtype) do { if ((dtype) == DT_F16) return hipErrorInvalidValue; } while (0)
template <typename T> static __device__ __forceinline__ uint4 ldv(const void* p, size_t vec_idx) {
  return ((const uint4*)p)[vec_idx];
}
template <typename T> static __device__ __forceinline__ void stv(void* p, size_t vec_idx, const uint4& v) {
  ((uint4*)p)[vec_idx] = v;
}

static inline int grid_for(size_t n, int block, int cap = 8192) {
  size_t g = (n + block - 1) / block;
  if (g > (size_t)cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace octseg
