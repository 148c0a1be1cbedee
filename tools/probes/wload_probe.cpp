// How fast do the 32 CUs of an XCD get the SAME 288 KiB out of their L2, and does the order in which they ask matter?  (conv_wstat.hip's
// prologue: one workgroup per CU, four waves, each wave loads its own 72 KiB of a layer's packed weights as 72 x 1-KiB wave loads.)
//   variant 0: every workgroup asks for fragment 0, 1, 2, ... (what the kernel does)
//   variant 1: workgroup c starts at fragment 9 * ((c / 8) % 8) of its 72 and wraps -- the CUs of an XCD walk different addresses at a time
//   variant 2: every workgroup has a COPY of its own (no sharing at all: 256 x 288 KiB = 72 MiB) -- the L2 / fabric rate without sharing
//   variant 3: one copy per XCD-local CU index mod 4 (four copies)
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 -x hip tools/probes/wload_probe.cpp -o tools/_build/wload_probe && tools/_build/wload_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int VAR>
__global__ __launch_bounds__(256, 1) void probe(const char* __restrict__ w, long long* __restrict__ cyc, float* __restrict__ sink) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, c = blockIdx.x;
  const size_t copy = VAR == 2 ? (size_t)c * 294912 : VAR == 3 ? (size_t)((c / 8) & 3) * 294912 : 0;
  const char* base = w + copy + (size_t)wid * 73728 + lane * 16;
  const int rot = VAR == 1 ? 9 * ((c / 8) & 7) : 0;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  f32x4 r[72];
#pragma unroll
  for (int k = 0; k < 72; ++k) {
    int f = k + rot; if (f >= 72) f -= 72;
    r[k] = *reinterpret_cast<const f32x4*>(base + (size_t)f * 1024);
  }
#pragma unroll
  for (int k = 0; k < 72; ++k) acc += r[k];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[c] = t1 - t0;
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[0] = acc.x;
}
int main() {
  char* w; long long* cyc; float* sink;
  hipMalloc(&w, (size_t)256 * 294912 + 4096); hipMalloc(&cyc, 256 * 8); hipMalloc(&sink, 64);
  hipMemset(w, 0, (size_t)256 * 294912);
  std::vector<long long> h(256);
  auto run = [&](int var) {
    for (int rep = 0; rep < 4; ++rep) {
      // a fresh launch with cold L2 for the weights would be the first launch of a layer: flush by touching 64 MiB elsewhere is not
      // needed -- the kernel's case is weights written by the pack kernel long before (L2 or MALL, not L1)
      if (var == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(256), 0, 0, w, cyc, sink);
      if (var == 1) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(256), 0, 0, w, cyc, sink);
      if (var == 2) hipLaunchKernelGGL(probe<2>, dim3(256), dim3(256), 0, 0, w, cyc, sink);
      if (var == 3) hipLaunchKernelGGL(probe<3>, dim3(256), dim3(256), 0, 0, w, cyc, sink);
      hipDeviceSynchronize();
      hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
      long long s = 0, mx = 0, mn = 1LL << 60;
      for (auto v : h) { s += v; if (v > mx) mx = v; if (v < mn) mn = v; }
      printf("{\"variant\": %d, \"rep\": %d, \"memtime_ticks_mean\": %.0f, \"min\": %lld, \"max\": %lld, \"bytes_per_wg\": 294912}\n", var, rep, (double)s / 256, mn, mx);
    }
  };
  for (int v = 0; v < 4; ++v) run(v);
  return 0;
}
