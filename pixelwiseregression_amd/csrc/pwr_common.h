// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of the PixelwiseRegression hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PWR_WAVE 64

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// Experiment switches exist only in the DEBUG build (-DPWR_DEBUG_BUILD: tools/build_debug.py -> tools/_build/libpwr_hip_dbg.so, loaded
// by tools/dbglib.py).  The shipped library has ONE configuration: in it PWR_DBG_ENV("X", d) is the constant d, no environment
// variable is read, and the branches of the other values fold away.  Measured negative results are recorded in DESIGN.md.
#ifdef PWR_DEBUG_BUILD
#include <cstdlib>
static inline int pwr_dbg_env_(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#define PWR_DBG_ENV(name, dflt) pwr_dbg_env_(name, dflt)
#else
#define PWR_DBG_ENV(name, dflt) (dflt)
#endif

// dtype tags of the C ABI (include/pwr.h)
#define PWR_F32 0
#define PWR_BF16 1

namespace pwr {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Block-wide sum of NV values per thread (all threads get the result). `red` needs NV*NWAVES floats.
template <int NV, int NWAVES>
__device__ __forceinline__ void block_sum(float (&v)[NV], float* red) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
  __syncthreads();  // protect `red` from a previous use
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) red[i * NWAVES + wid] = v[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NWAVES; ++w) s += red[i * NWAVES + w];
    v[i] = s;
  }
}
// block-wide sum of one value, 256 threads (4 waves); `red` needs 4 floats
__device__ __forceinline__ float block_sum1(float v, float* red) {
  float a[1] = {v};
  block_sum<1, 4>(a, red);
  return a[0];
}
template <int NWAVES>
__device__ __forceinline__ float block_max(float v, float* red) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  v = wave_max(v);
  __syncthreads();
  if (lane == 0) red[wid] = v;
  __syncthreads();
  float s = red[0];
#pragma unroll
  for (int w = 1; w < NWAVES; ++w) s = fmaxf(s, red[w]);
  return s;
}

// ---- element type traits: activations are stored as float or bf16 ("T"), math is fp32 ----
template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int kPer16B = 4;
  __device__ static __forceinline__ float to_f(float v) { return v; }
  __device__ static __forceinline__ float from_f(float v) { return v; }
};
template <> struct Elem<bf16_t> {
  static constexpr int kPer16B = 8;
  __device__ static __forceinline__ float to_f(bf16_t v) { return (float)v; }
  __device__ static __forceinline__ bf16_t from_f(float v) { return (bf16_t)v; }
};

// 16-byte vector of T
template <typename T> struct Vec16;
template <> struct Vec16<float> { typedef f32x4 type; };
template <> struct Vec16<bf16_t> { typedef bf16x8 type; };

__host__ __device__ __forceinline__ int ceil_div(int a, int b) { return (a + b - 1) / b; }

}  // namespace pwr
