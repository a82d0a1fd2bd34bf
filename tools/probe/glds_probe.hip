// Probe: semantics of __builtin_amdgcn_global_load_lds (16-byte, lane-linear LDS destination) on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned* src, unsigned* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // each wave copies 1 KiB: global per-lane address, LDS wave-uniform base
  const char* g = (const char*)src + wave * 1024 + lane * 16;
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)(smem + wave * 1024), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int i = tid; i < 1024; i += 256) out[i] = ((const unsigned*)smem)[i];
}
int main() {
  std::vector<unsigned> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = i * 3 + 1;
  unsigned *d, *o;
  hipMalloc(&d, 4096); hipMalloc(&o, 4096);
  hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 4096, 0, d, o);
  std::vector<unsigned> r(1024);
  hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 1024; ++i) bad += r[i] != h[i];
  printf("glds probe: %d mismatches (%s)\n", bad, hipGetErrorString(hipGetLastError()));
  return bad != 0;
}
