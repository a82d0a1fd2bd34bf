// hip_host_stubs.cpp -- a recording stand-in for the handful of HIP runtime entry points the HOST half of liboctseg_hip.so calls, so that
// tools/plan_dryrun.cpp can drive the planner and the forward / backward executors under AddressSanitizer + UBSan on a machine with no
// GPU and no HIP runtime in the process.  Test infrastructure only (linked into build/asan/plan_dryrun, never into the library).
//
// Nothing is executed: a "launch" checks its geometry against what gfx950 accepts (non-empty grid, block of 64..1024 threads in whole
// waves, dynamic LDS <= 160 KiB) and is counted; memsets / copies check that their device range lies inside a range the test registered
// (dry_register_range) -- the fake arenas are address ranges that are never dereferenced.
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {
struct Range { const char* lo; const char* hi; };
std::vector<Range> g_ranges;
unsigned long long g_launches = 0, g_memsets = 0, g_copies = 0, g_events = 0, g_errors = 0;
dim3 g_grid, g_block; size_t g_shmem = 0; hipStream_t g_stream = nullptr;
size_t g_max_dyn_lds = 64 * 1024;   // what a kernel may use before hipFuncSetAttribute raises it (tracked per process, conservatively)
int g_stream_ids = 0;

bool in_ranges(const void* p, size_t n) {
  const char* c = (const char*)p;
  for (auto& r : g_ranges) if (c >= r.lo && c + n <= r.hi) return true;
  return false;
}
void complain(const char* what) { fprintf(stderr, "hip_host_stubs: %s\n", what); ++g_errors; }
}  // namespace

extern "C" {
void dry_register_range(const void* lo, size_t bytes) { g_ranges.push_back({(const char*)lo, (const char*)lo + bytes}); }
void dry_clear_ranges() { g_ranges.clear(); }
unsigned long long dry_launches() { return g_launches; }
unsigned long long dry_memops() { return g_memsets + g_copies; }
unsigned long long dry_errors() { return g_errors; }

// ---- registration hooks emitted by clang for every translation unit with kernels
void** __hipRegisterFatBinary(const void*) { static void* h = nullptr; return &h; }
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
void __hipUnregisterFatBinary(void**) {}
hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t st) {
  g_grid = grid; g_block = block; g_shmem = shmem; g_stream = st;
  return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3* grid, dim3* block, size_t* shmem, hipStream_t* st) {
  *grid = g_grid; *block = g_block; *shmem = g_shmem; *st = g_stream;
  return hipSuccess;
}
hipError_t hipLaunchKernel(const void* fn, dim3 grid, dim3 block, void** args, size_t shmem, hipStream_t) {
  ++g_launches;
  const unsigned long long threads = (unsigned long long)block.x * block.y * block.z;
  if (!fn || !args) complain("launch without a function or an argument list");
  if (grid.x == 0 || grid.y == 0 || grid.z == 0) complain("empty grid");
  if (grid.y > 65535 || grid.z > 65535) complain("grid.y / grid.z beyond 65535");
  if (threads == 0 || threads > 1024 || threads % 64 != 0) complain("block is not 1..16 whole waves");
  if (shmem > 160 * 1024) complain("more than 160 KiB of dynamic LDS");
  if (shmem > g_max_dyn_lds) complain("dynamic LDS above 64 KiB without hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
  return hipSuccess;
}
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute attr, int value) {
  if (attr == hipFuncAttributeMaxDynamicSharedMemorySize) {
    if (value > 160 * 1024) complain("MaxDynamicSharedMemorySize beyond the 160 KiB of a CU");
    if ((size_t)value > g_max_dyn_lds) g_max_dyn_lds = (size_t)value;
  }
  return hipSuccess;
}
hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "stub"; }
hipError_t hipMemsetAsync(void* dst, int, size_t n, hipStream_t) {
  ++g_memsets;
  if (!in_ranges(dst, n)) complain("hipMemsetAsync outside every registered device range");
  return hipSuccess;
}
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind kind, hipStream_t) {
  ++g_copies;
  if (kind == hipMemcpyHostToDevice) {
    if (!in_ranges(dst, n)) complain("H2D copy outside every registered device range");
    volatile unsigned char acc = 0;                      // touch the host source: ASan checks it is readable end to end
    for (size_t i = 0; i < n; ++i) acc ^= ((const unsigned char*)src)[i];
    (void)acc;
  } else if (!in_ranges(dst, n) && !in_ranges(src, n)) complain("copy touches no registered device range");
  return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t* st, unsigned) { *st = (hipStream_t)(uintptr_t)(0x5000 + 16 * ++g_stream_ids); return hipSuccess; }
hipError_t hipStreamCreateWithPriority(hipStream_t* st, unsigned f, int) { return hipStreamCreateWithFlags(st, f); }
hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = -1; return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)(uintptr_t)(0x9000 + 16 * ++g_events); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { if (!e) complain("record of a null event"); return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 1.0f; return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t e, unsigned) { if (!e) complain("wait for a null event"); return hipSuccess; }
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipSuccess; }
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = (hipGraph_t)(uintptr_t)0xA000; return hipSuccess; }
hipError_t hipGraphInstantiate(hipGraphExec_t* ex, hipGraph_t, hipGraphNode_t*, char*, size_t) { *ex = (hipGraphExec_t)(uintptr_t)0xB000; return hipSuccess; }
hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipSuccess; }
hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
}
