// Small entry points that belong to no kernel family.
#include <hip/hip_runtime.h>

#include "pwr.h"
#ifdef PWR_DEBUG_BUILD
#include "pwr_debug.h"
#endif

extern "C" int pwr_abi_version(void) { return PWR_ABI_VERSION; }

#ifdef PWR_DEBUG_BUILD
namespace {
// dst = src if *flag != 0 (grid-stride, 16-byte vectors); every workgroup leaves at once otherwise
__global__ __launch_bounds__(256) void copy_if_kernel(const int* __restrict__ flag, const uint4* __restrict__ src, uint4* __restrict__ dst,
                                                      size_t n16) {
  if (*flag == 0) return;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
}  // namespace

// Debugging aid (tools/race_hunt.py): copy `bytes` (a multiple of 16) from src to dst on the device iff *flag != 0, without a host
// round trip -- a snapshot of the arena is taken in the very step whose gradient differs, at the cost of one empty launch otherwise.
extern "C" int pwr_debug_copy_if(const int* flag, const void* src, void* dst, size_t bytes, void* stream) {
  if (bytes % 16) return PWR_EINVAL;
  hipLaunchKernelGGL(copy_if_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, flag, (const uint4*)src, (uint4*)dst, bytes / 16);
  return (int)hipGetLastError();
}
#endif   // PWR_DEBUG_BUILD
