// Shared by the implicit-GEMM convolution kernels (conv_mfma.hip: universal; conv_patch.hip: LDS-resident input patch).
#pragma once
#include "pwr_common.h"

namespace pwr {

struct ConvParams {
  const void* x;          // [B,H,W,Cin] T   (forward: input; dgrad: dy)
  const void* w;          // packed weights, see pack_weights kernel
  const float* bias;      // [Cout] or null
  const float* in_norm;   // [4][B][Cin] = mean, rstd, scale, beta of the input's norm (or null): NR prologue
  const void* residual;   // [B,Ho,Wo,Cout] T or null (added in the epilogue)
  void* y;                // [B,Ho,Wo,Cout] T or null
  float* y_nchw;          // [B,Cout,Ho,Wo] fp32 or null
  int B, H, W, Cin, Ho, Wo, Cout, CoutPad;
  int ksize, stride, pad, mode, relu_in, KCH, M;
};

struct WgradParams {
  const void* x;          // [B,H,W,Cin] T  forward input (pre-NR)
  const void* dy;         // [B,Ho,Wo,Cout] T
  const float* in_norm;   // NR prologue of the forward conv (or null), [4][B][Cin]
  float* slab;            // [S][taps][CinPad128][CoutPad] fp32 partials
  int B, H, W, Cin, Ho, Wo, Cout, CoutPad, CinPad;
  int ksize, stride, pad, relu_in, M, S, steps_per_split;
};

template <typename T> struct Mma;
template <> struct Mma<bf16_t> { static constexpr int KE = 32; static constexpr int EP = 8; };
template <> struct Mma<float> { static constexpr int KE = 16; static constexpr int EP = 4; };

__device__ __forceinline__ int lds_off(int row, int slot) { return row * 64 + (((slot ^ (row >> 2)) & 3) << 4); }

template <typename T, int MR, int NR>
__device__ __forceinline__ void mma_tile(const char* lA, const char* lB, int a_row0, int b_row0, int lane,
                                         f32x16 (&acc)[MR][NR]) {
  typedef typename Vec16<T>::type V;
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int ss = 0; ss < 2; ++ss) {
    V a[MR], b[NR];
#pragma unroll
    for (int i = 0; i < MR; ++i) a[i] = *reinterpret_cast<const V*>(lA + lds_off(a_row0 + i * 32 + r, 2 * ss + h));
#pragma unroll
    for (int j = 0; j < NR; ++j) b[j] = *reinterpret_cast<const V*>(lB + lds_off(b_row0 + j * 32 + r, 2 * ss + h));
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        if constexpr (sizeof(T) == 2) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
        }
      }
  }
}

// v = relu?((v - mean)*scale + beta) on one 16-byte vector; st points at mean[b][c0] of a [4][B][C] norm state
// (mean, rstd, scale = gamma*rstd, beta).  Subtracting the mean first keeps the cancellation exact-ish, like
// ATen's (x - mean) * invstd * gamma + beta.
template <typename T>
__device__ __forceinline__ typename Vec16<T>::type nr_transform(typename Vec16<T>::type v, const float* st, size_t plane,
                                                                int relu) {
  constexpr int EP = Mma<T>::EP;
  typename Vec16<T>::type o;
#pragma unroll
  for (int e = 0; e < EP; ++e) {
    float f = fmaf(Elem<T>::to_f(v[e]) - st[e], st[2 * plane + e], st[3 * plane + e]);
    if (relu) f = fmaxf(f, 0.f);
    o[e] = Elem<T>::from_f(f);
  }
  return o;
}

__device__ __forceinline__ int xcd_remap(int bid, int n) {
  // give each XCD (blocks b, b+8, ... share one) a contiguous range of tiles: neighbours share halo rows in L2
  const int q = n >> 3, r = n & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}


static inline int pick_bn(int cout) { return cout > 64 ? 128 : (cout > 32 ? 64 : 32); }

// conv_patch.hip
bool conv_patch_applicable(const ConvParams& p, int dtype);
int launch_conv_patch(const ConvParams& p, int dtype, hipStream_t s);

}  // namespace pwr
