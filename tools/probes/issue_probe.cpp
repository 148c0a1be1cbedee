// Issue-rate probe for ONE wave per SIMD on gfx950 (standalone; tools/issue_probe.sh builds and runs it): what else fits beside a stream of
// v_mfma_f32_32x32x16_bf16 when the wave that issues the MFMAs also has to issue everything else -- the regime of conv_wstat.hip.
// Every instruction is a volatile asm statement, so the emitted order is the written order.  Output: shader cycles (s_memtime) per MFMA slot.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// MODE bits: 1 ds_read_b128 per slot (ring of 8, consumed by the MFMA 8 slots later), 2 accumulators in AGPRs, 4 s_waitcnt lgkmcnt(7) per slot,
// 8 the extra instructions are SALU (s_add_u32), 16 they are s_nop 0, 32 they form ONE dependent chain (else round-robin over 8 registers),
// 64 they are v_cvt_pk_bf16_f32, 128 a single accumulator chain (every MFMA depends on the one before), 256 v_pk_add_f32, 512 v_med3_f32
template <int NV, int MODE>
__global__ __launch_bounds__(256, 1) void probe(long long* out, float* sink, int iters) {
  __shared__ __attribute__((aligned(16))) char smem[120 * 1024];      // one workgroup per CU
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 120 * 256; i += 256) reinterpret_cast<float*>(smem)[i] = 1.0f;
  __syncthreads();
  f32x16 acc0 = {}, acc1 = {};
  u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  u32x4 pf[8];
  for (int i = 0; i < 8; ++i) pf[i] = a;
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = 1.0f + i + lane;
  const float c1 = 0.999f, c2 = 0.001f;
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  f32x2 xp[4], cp = {0.5f, 0.25f};
  for (int i = 0; i < 4; ++i) xp[i] = f32x2{1.0f + lane, 2.0f + i};
  unsigned addr = (unsigned)(size_t)(smem) + lane * 272;
  unsigned s0 = 1;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (MODE & 4) asm volatile("s_waitcnt lgkmcnt(7)");
      if (MODE & 2) {
        if ((s & 1) && !(MODE & 128)) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc1) : "v"(a), "v"(pf[s]));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc0) : "v"(a), "v"(pf[s]));
      } else {
        if ((s & 1) && !(MODE & 128)) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(pf[s]));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(pf[s]));
      }
      if (MODE & 1) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(pf[s]) : "v"(addr), "n"(s * 64));
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        if (MODE & 8) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s0) : : "scc");
        else if (MODE & 256) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(xp[v & 3]) : "v"(cp));
        else if (MODE & 512) asm volatile("v_med3_f32 %0, %0, 0, %1" : "+v"(x[(MODE & 32) ? 0 : v & 7]) : "v"(c1));
        else if (MODE & 16) asm volatile("s_nop 0");
        else if (MODE & 64) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x[(MODE & 32) ? 0 : v & 7]) : "v"(c1));
        else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[(MODE & 32) ? 0 : v & 7]) : "v"(c1), "v"(c2));
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)");
  const long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0.f;
  for (int i = 0; i < 16; ++i) r += acc0[i] + acc1[i];
  for (int i = 0; i < 8; ++i) r += x[i] + __builtin_bit_cast(float, pf[i][0]);
  for (int i = 0; i < 4; ++i) r += xp[i][0] + xp[i][1];
  sink[blockIdx.x * 256 + threadIdx.x] = r + (float)s0;
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int NV, int MODE>
static void run(const char* what, long long* d_out, float* d_sink, int nwg) {
  const int iters = 2000;
  hipLaunchKernelGGL((probe<NV, MODE>), dim3(nwg), dim3(256), 0, 0, d_out, d_sink, 10);
  hipLaunchKernelGGL((probe<NV, MODE>), dim3(nwg), dim3(256), 0, 0, d_out, d_sink, iters);
  hipDeviceSynchronize();
  std::vector<long long> h(nwg);
  hipMemcpy(h.data(), d_out, nwg * sizeof(long long), hipMemcpyDeviceToHost);
  double mean = 0; long long mn = h[0], mx = h[0];
  for (auto v : h) { mean += v; mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
  mean /= nwg;
  printf("{\"what\": \"%s\", \"extra_per_slot\": %d, \"mode\": %d, \"cycles_per_slot\": %.2f, \"min\": %.2f, \"max\": %.2f}\n", what, NV, MODE,
         mean / (iters * 8.0), mn / (iters * 8.0), mx / (iters * 8.0));
}

#define SWEEP(MODE, what) \
  run<0, MODE>(what, d_out, d_sink, nwg); run<1, MODE>(what, d_out, d_sink, nwg); run<2, MODE>(what, d_out, d_sink, nwg); \
  run<3, MODE>(what, d_out, d_sink, nwg); run<4, MODE>(what, d_out, d_sink, nwg); run<5, MODE>(what, d_out, d_sink, nwg); \
  run<6, MODE>(what, d_out, d_sink, nwg); run<8, MODE>(what, d_out, d_sink, nwg);

int main(int argc, char** argv) {
  const int nwg = argc > 1 ? atoi(argv[1]) : 256;
  long long* d_out; float* d_sink;
  hipMalloc(&d_out, nwg * sizeof(long long));
  hipMalloc(&d_sink, nwg * 256 * sizeof(float));
  SWEEP(0, "mfma (acc in VGPRs) + n x v_fma_f32 on independent registers")
  SWEEP(2, "mfma (acc in AGPRs) + n x v_fma_f32 on independent registers")
  SWEEP(32, "mfma (VGPR acc) + n x v_fma_f32, one dependent chain")
  SWEEP(1 | 4, "mfma (VGPR acc) + ds_read_b128 + s_waitcnt + n x v_fma_f32")
  SWEEP(1 | 4 | 2, "mfma (AGPR acc) + ds_read_b128 + s_waitcnt + n x v_fma_f32")
  SWEEP(1 | 4 | 8, "mfma (VGPR acc) + ds_read_b128 + s_waitcnt + n x s_add_u32")
  SWEEP(1 | 4 | 16, "mfma (VGPR acc) + ds_read_b128 + s_waitcnt + n x s_nop")
  SWEEP(1 | 4 | 64, "mfma (VGPR acc) + ds_read_b128 + s_waitcnt + n x v_cvt_pk_bf16_f32")
  SWEEP(1 | 4 | 256, "mfma (VGPR acc) + ds_read_b128 + s_waitcnt + n x v_pk_add_f32")
  SWEEP(1 | 4 | 512, "mfma (VGPR acc) + ds_read_b128 + s_waitcnt + n x v_med3_f32")
  SWEEP(128, "mfma, ONE accumulator chain (VGPR) + n x v_fma_f32")
  SWEEP(128 | 2, "mfma, ONE accumulator chain (AGPR) + n x v_fma_f32")
  return 0;
}
