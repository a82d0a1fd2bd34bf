// Probe 2: inline-asm global_load_lds_dwordx4 with M0 as the LDS base, at a high (>64 KiB) LDS offset,
// issued while compiler-visible ds_reads run; completion by a hand-counted s_waitcnt vmcnt.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__global__ void k(const unsigned* src, unsigned* out, int hi_off) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;  // LDS byte address of smem
  for (int i = tid; i < 1024; i += 256) ((unsigned*)smem)[i] = 7u;   // low region: something to ds_read meanwhile
  __syncthreads();
  const char* g = (const char*)src + wave * 1024 + lane * 16;
  dma16(g, base + hi_off + wave * 1024);
  unsigned acc = 0;
  for (int i = 0; i < 8; ++i) acc += ((const unsigned*)smem)[(tid + i * 37) & 1023];  // unrelated LDS reads in flight
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  for (int i = tid; i < 1024; i += 256) out[i] = ((const unsigned*)(smem + hi_off))[i] + (acc == 56u ? 0u : 1000000u);
}
int main() {
  std::vector<unsigned> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = i * 3 + 1;
  unsigned *d, *o;
  (void)hipMalloc(&d, 4096); (void)hipMalloc(&o, 4096);
  (void)hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  int rc = 0;
  for (int hi : {4096, 60 * 1024, 100 * 1024, 150 * 1024}) {
    (void)hipMemset(o, 0, 4096);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), hi + 4096, 0, d, o, hi);
    std::vector<unsigned> r(1024);
    (void)hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; ++i) bad += r[i] != h[i];
    printf("asm glds at LDS offset %6d: %d mismatches (%s)\n", hi, bad, hipGetErrorString(hipGetLastError()));
    rc |= bad != 0;
  }
  return rc;
}
