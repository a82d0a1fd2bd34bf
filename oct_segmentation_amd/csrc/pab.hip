// pab.hip -- the position attention block of smp's MAnet decoder (decoders/manet/decoder.py PAB; reference sweep configs/tune.yaml:17 ``MAnet``
// through smp.create_model, src/models/smp/model.py:38-44), between its four convolutions (which run on the conv kernels):
//
//     S = center topT          [HW x HW] per image          (center, top: 1x1 convs to 64 channels)
//     P = softmax over ALL HW^2 entries of S                (upstream: view(bsize, -1) + Softmax(dim=1))
//     M = P bottom             [HW x C]                     (bottom: 3x3 conv, C channels)
//     y = x + reshape(M, [C, h, w])                         (upstream reshapes [HW, C] to [C, h, w] WITHOUT a transpose: kept)
//
// and the gradients of all of it.  On the deepest feature only (22 x 22 at 704^2: HW = 484), 1 GFLOP per image: small strided f32 GEMMs
// from LDS tiles, f32 scratch for S / P / M; no MFMA (activation x activation products, nothing to pack).
#include "common.h"
#include "ev.h"
#include "kernels.h"

namespace octseg {

template <typename T> static __device__ __forceinline__ float pab_ld(const void* p, size_t i, int is_f32) {
  if (is_f32 || sizeof(T) == 4) return ((const float*)p)[i];
  if (sizeof(T) == 2 && T::kHalf) return (float)__builtin_bit_cast(_Float16, ((const unsigned short*)p)[i]);
  return __uint_as_float((unsigned)((const unsigned short*)p)[i] << 16);
}
template <typename T> static __device__ __forceinline__ void pab_st(void* p, size_t i, float v, int is_f32, int accum) {
  if (is_f32 || sizeof(T) == 4) { float* q = (float*)p + i; *q = accum ? *q + v : v; return; }
  unsigned short* q = (unsigned short*)p + i;
  if (T::kHalf) {
    if (accum) v += (float)__builtin_bit_cast(_Float16, *q);
    const _Float16 h = (_Float16)v; *q = __builtin_bit_cast(unsigned short, h);
  } else {
    if (accum) v += __uint_as_float((unsigned)*q << 16);
    *q = (unsigned short)(pk_bf16(v, 0.f) & 0xffffu);
  }
}
struct PabF32 { static constexpr bool kHalf = false; float x; };
struct PabBf16 { static constexpr bool kHalf = false; unsigned short x; };
struct PabF16 { static constexpr bool kHalf = true; unsigned short x; };

// C[b][m][n] (+)= sum_k A[b][m][k] B[b][k][n], element (m, k) of A at sAm m + sAk k (likewise B, C); 16 x 16 tiles through LDS
template <typename T>
__global__ __launch_bounds__(256) void pab_gemm_kernel(const PabGemm g) {
  __shared__ float sa[16][17], sb[16][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int m = blockIdx.y * 16 + ty, n = blockIdx.x * 16 + tx;
  const size_t b = blockIdx.z;
  const char* A = (const char*)g.A + b * g.sAb * (g.a_f32 ? 4 : sizeof(T));
  const char* B = (const char*)g.B + b * g.sBb * (g.b_f32 ? 4 : sizeof(T));
  float acc = 0.f;
  for (int k0 = 0; k0 < g.K; k0 += 16) {
    const int ka = k0 + tx, kb = k0 + ty;
    sa[ty][tx] = (m < g.M && ka < g.K) ? pab_ld<T>(A, (size_t)m * g.sAm + (size_t)ka * g.sAk, g.a_f32) : 0.f;
    sb[ty][tx] = (kb < g.K && n < g.N) ? pab_ld<T>(B, (size_t)kb * g.sBk + (size_t)n * g.sBn, g.b_f32) : 0.f;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) acc = fmaf(sa[ty][k], sb[k][tx], acc);
    __syncthreads();
  }
  if (m < g.M && n < g.N) {
    char* Cp = (char*)g.C + b * g.sCb * (g.c_f32 ? 4 : sizeof(T));
    pab_st<T>(Cp, (size_t)m * g.sCm + (size_t)n * g.sCn, acc, g.c_f32, g.accum);
  }
}
hipError_t launch_pab_gemm(int dtype, const PabGemm& g, hipStream_t st) {
  const dim3 grid((g.N + 15) / 16, (g.M + 15) / 16, g.batch);
  if (dtype == DT_F32) hipLaunchKernelGGL(pab_gemm_kernel<PabF32>, grid, dim3(256), 0, st, g);
  else if (dtype == DT_F16) hipLaunchKernelGGL(pab_gemm_kernel<PabF16>, grid, dim3(256), 0, st, g);
  else hipLaunchKernelGGL(pab_gemm_kernel<PabBf16>, grid, dim3(256), 0, st, g);
  return hipGetLastError();
}

// softmax over ALL n entries of an image's map (forward, in place), or its gradient dS = P (dP - sum(P dP)) written over dP; one workgroup
// of 1024 threads per image, sums in double in a fixed order
__global__ __launch_bounds__(1024) void pab_softmax_kernel(float* S, const float* P, size_t n, int backward) {
  __shared__ double red[1024];
  __shared__ float redf[1024];
  float* s = S + (size_t)blockIdx.x * n;
  const int t = threadIdx.x;
  if (!backward) {
    float mx = -__builtin_inff();
    for (size_t i = t; i < n; i += 1024) mx = fmaxf(mx, s[i]);
    redf[t] = mx;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) { if (t < o) redf[t] = fmaxf(redf[t], redf[t + o]); __syncthreads(); }
    mx = redf[0];
    double sum = 0.0;
    for (size_t i = t; i < n; i += 1024) { const float e = expf(s[i] - mx); s[i] = e; sum += (double)e; }
    red[t] = sum;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) { if (t < o) red[t] += red[t + o]; __syncthreads(); }
    const float inv = (float)(1.0 / red[0]);
    for (size_t i = t; i < n; i += 1024) s[i] *= inv;
  } else {
    const float* p = P + (size_t)blockIdx.x * n;
    double dot = 0.0;
    for (size_t i = t; i < n; i += 1024) dot += (double)p[i] * (double)s[i];
    red[t] = dot;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) { if (t < o) red[t] += red[t + o]; __syncthreads(); }
    const float d = (float)red[0];
    for (size_t i = t; i < n; i += 1024) s[i] = p[i] * (s[i] - d);
  }
}
hipError_t launch_pab_softmax(float* S, const float* P, int batch, size_t n, int backward, hipStream_t st) {
  hipLaunchKernelGGL(pab_softmax_kernel, dim3(batch), dim3(1024), 0, st, S, P, n, backward);
  return hipGetLastError();
}

// the un-transposed reshape: flat index f of M [HW][C] is element (channel f / HW, pixel f % HW) of the NCHW map that is added to x.
// forward: y[p][ch] = x[p][ch] + M[f];   backward: dM[f] = dy[p][ch]      (NHWC tensors x, y, dy)
template <typename T>
__global__ __launch_bounds__(256) void pab_mix_kernel(const void* x, const float* M, void* y, float* dM, const void* dy, int HW, int C, size_t total) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const size_t per = (size_t)HW * C;
    const size_t n = e / per, f = e - n * per;
    const size_t ch = f / HW, p = f - ch * HW;
    const size_t o = n * per + p * C + ch;
    if (dM == nullptr) pab_st<T>(y, o, pab_ld<T>(x, o, 0) + M[e], 0, 0);
    else dM[e] = pab_ld<T>(dy, o, 0);
  }
}
hipError_t launch_pab_mix(int dtype, const void* x, const float* M, void* y, float* dM, const void* dy, int N, int HW, int C, hipStream_t st) {
  const size_t total = (size_t)N * HW * C;
  const dim3 grid(grid_for(total, 256));
  if (dtype == DT_F32) hipLaunchKernelGGL(pab_mix_kernel<PabF32>, grid, dim3(256), 0, st, x, M, y, dM, dy, HW, C, total);
  else if (dtype == DT_F16) hipLaunchKernelGGL(pab_mix_kernel<PabF16>, grid, dim3(256), 0, st, x, M, y, dM, dy, HW, C, total);
  else hipLaunchKernelGGL(pab_mix_kernel<PabBf16>, grid, dim3(256), 0, st, x, M, y, dM, dy, HW, C, total);
  return hipGetLastError();
}

}  // namespace octseg
