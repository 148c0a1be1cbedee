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

// The value of lane ^ O (O a power of two below 64) WITHOUT the LDS crossbar (round 6).  __shfl_xor compiles to ds_bpermute_b32: an LDS-pipe
// round trip per value and butterfly step, which the latency-bound kernels of the chain (one-launch ResBlocks, conv epilogue statistics,
// decoder reductions) pay in full -- they have no other wave to hide it behind.  O = 1, 2: DPP quad permutes; 4: row_shl:4 / row_shr:4 under
// bank masks; 8: row_ror:8; 16 / 32: v_permlane16_swap / v_permlane32_swap of the value with itself (the swap leaves the pair's lower and
// upper value in its two results; the partner's is the one that is not the lane's own).  Butterflies built on it add or max self and
// partner exactly like `v op= __shfl_xor(v, O)`: both operations commute, the bits are the same.
template <int O>
__device__ __forceinline__ float lane_xor(float x) {
  const int u = __builtin_bit_cast(int, x);
  if constexpr (O == 1) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, u, 0xB1 /* quad_perm:[1,0,3,2] */, 0xf, 0xf, false));
  else if constexpr (O == 2) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, u, 0x4E /* quad_perm:[2,3,0,1] */, 0xf, 0xf, false));
  else if constexpr (O == 4) {
    int t = __builtin_amdgcn_update_dpp(0, u, 0x104 /* row_shl:4 */, 0xf, 0x5, false);       // lanes 0-3 and 8-11 of a row read lane + 4
    t = __builtin_amdgcn_update_dpp(t, u, 0x114 /* row_shr:4 */, 0xf, 0xA, false);           // lanes 4-7 and 12-15 read lane - 4
    return __builtin_bit_cast(float, t);
  } else if constexpr (O == 8) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, u, 0x128 /* row_ror:8 */, 0xf, 0xf, false));
  else if constexpr (O == 16) {
    const auto r = __builtin_amdgcn_permlane16_swap((unsigned)u, (unsigned)u, false, false);
    return __builtin_bit_cast(float, (__builtin_amdgcn_workitem_id_x() & 16) ? (unsigned)r[0] : (unsigned)r[1]);
  } else {
    static_assert(O == 32, "butterfly step");
    const auto r = __builtin_amdgcn_permlane32_swap((unsigned)u, (unsigned)u, false, false);
    return __builtin_bit_cast(float, (__builtin_amdgcn_workitem_id_x() & 32) ? (unsigned)r[0] : (unsigned)r[1]);
  }
}
// x + (x of lane ^ O): for 16 / 32 the swap's two results ARE self and partner, whichever lane: their sum needs no select
template <int O>
__device__ __forceinline__ float lane_xor_add(float x) {
  if constexpr (O == 16) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
  } else if constexpr (O == 32) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
  } else return x + lane_xor<O>(x);
}

__device__ __forceinline__ float wave_sum(float v) {
  v = lane_xor_add<32>(v); v = lane_xor_add<16>(v); v = lane_xor_add<8>(v);
  v = lane_xor_add<4>(v); v = lane_xor_add<2>(v); v = lane_xor_add<1>(v);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, lane_xor<32>(v)); v = fmaxf(v, lane_xor<16>(v)); v = fmaxf(v, lane_xor<8>(v));
  v = fmaxf(v, lane_xor<4>(v)); v = fmaxf(v, lane_xor<2>(v)); v = fmaxf(v, lane_xor<1>(v));
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
