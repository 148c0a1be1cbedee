// InstanceNorm2d / BatchNorm2d (affine) + ReLU, forward statistics and backward, on NHWC activations.
//
// Replaces the `norm(...)` + ReLU pairs of /root/reference/model.py (stem :166-185, ResBlock :11-18, heads
// :55-63/:104-112).  The normalised tensor is never written: the forward only produces per-(b,c)
// scale = gamma*rstd and shift = beta - mean*scale, which the consuming conv applies while staging its A
// operand (conv_mfma.hip, "NR prologue").  HBM traffic: forward reads y once; backward reads (g, y) once
// for the two reductions and once more for the element-wise apply, writes dy once.
//
// Statistics use shifted sums (shift k[c] = first pixel of the sample / batch) so that
// var = E[(x-k)^2] - E[x-k]^2 does not cancel catastrophically; partials per pixel chunk are written to a
// slab and combined in a fixed order (deterministic, no atomics).
#include <cstdlib>

#include "pwr_common.h"
#include "pwr.h"

namespace pwr {

// partial[((b*nchunks + chunk)*2 + {0,1})*C + c]
template <typename T>
__global__ __launch_bounds__(256) void norm_partial_kernel(const T* __restrict__ y, float* __restrict__ partial, int HW,
                                                           int C, int nchunks, int batch_mode) {
  constexpr int EP = Elem<T>::kPer16B;
  typedef typename Vec16<T>::type V;
  extern __shared__ float red[];  // [pl][2][C]
  const int chunk = blockIdx.x, b = blockIdx.y;
  const int cpp = C / EP, pl = 256 / cpp;
  const int cq = threadIdx.x % cpp, pj = threadIdx.x / cpp;
  const int per = (HW + nchunks - 1) / nchunks;
  const int p0 = chunk * per, p1 = min(HW, p0 + per);
  const T* base = y + (size_t)b * HW * C;
  const T* kb = batch_mode ? y : base;   // shift source: pixel 0 of sample b (instance) / of sample 0 (batch)
  float s1[EP], s2[EP], k[EP];
#pragma unroll
  for (int e = 0; e < EP; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  if (pj < pl) {
    V kv = *reinterpret_cast<const V*>(kb + cq * EP);
#pragma unroll
    for (int e = 0; e < EP; ++e) k[e] = Elem<T>::to_f(kv[e]);
#pragma unroll 4
    for (int pp = p0 + pj; pp < p1; pp += pl) {
      V v = *reinterpret_cast<const V*>(base + (size_t)pp * C + cq * EP);
#pragma unroll
      for (int e = 0; e < EP; ++e) {
        const float d = Elem<T>::to_f(v[e]) - k[e];
        s1[e] += d;
        s2[e] = fmaf(d, d, s2[e]);
      }
    }
#pragma unroll
    for (int e = 0; e < EP; ++e) {
      red[(pj * 2 + 0) * C + cq * EP + e] = s1[e];
      red[(pj * 2 + 1) * C + cq * EP + e] = s2[e];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    float s = 0.f;
    for (int j = 0; j < pl; ++j) s += red[j * 2 * C + i];
    partial[((size_t)(b * nchunks + chunk) * 2) * C + i] = s;
  }
}

// norm_partial_kernel with the tensor's PRODUCER fused in (round 6, instance norm): the input of a ResBlock's first norm on the 64x64 /
// 32x32 hourglass levels is the 2x2 max-pool of the level above (model.py:40) or the nearest up-sample of the level below + the skip
// (model.py:45-47); each was a launch of its own in front of the statistics launch.  SRC 1: y = maxpool2x2(xa [B,2H,2W,C]); SRC 2:
// y = xa [B,H,W,C] + up(xh [B,H/2,W/2,C]).  The value is computed with maxpool_fwd_kernel's / upsample_add_kernel's arithmetic, WRITTEN to
// y (the convs and the backward pass read it) and accumulated -- same thread -> (pixel, channel slot) map, same order as norm_partial_kernel:
// the partials, the finalised state and y are the bytes the two launches gave.
template <typename T, int SRC>
__device__ __forceinline__ typename Vec16<T>::type norm_src_value(const T* __restrict__ xa, const T* __restrict__ xh, int b, int pp, int H, int W, int C,
                                                                   int cq) {
  constexpr int EP = Elem<T>::kPer16B;
  typedef typename Vec16<T>::type V;
  const int oy = pp / W, ox = pp - oy * W;
  V o;
  if constexpr (SRC == 1) {
    const T* p = xa + (((size_t)b * 2 * H + 2 * oy) * 2 * W + 2 * ox) * C + cq * EP;
    const V v00 = *reinterpret_cast<const V*>(p), v01 = *reinterpret_cast<const V*>(p + C);
    const V v10 = *reinterpret_cast<const V*>(p + (size_t)2 * W * C), v11 = *reinterpret_cast<const V*>(p + (size_t)2 * W * C + C);
#pragma unroll
    for (int e = 0; e < EP; ++e)
      o[e] = Elem<T>::from_f(fmaxf(fmaxf(Elem<T>::to_f(v00[e]), Elem<T>::to_f(v01[e])), fmaxf(Elem<T>::to_f(v10[e]), Elem<T>::to_f(v11[e]))));
  } else {
    // (exact 2x up-sample: ATen's nearest source index floor(dst * 0.5f) = dst >> 1, as pool.hip's nearest_src gives for in = out / 2)
    const V hv = *reinterpret_cast<const V*>(xh + (((size_t)b * (H / 2) + (oy >> 1)) * (W / 2) + (ox >> 1)) * C + cq * EP);
    const V sv = *reinterpret_cast<const V*>(xa + ((size_t)b * H * W + pp) * C + cq * EP);
#pragma unroll
    for (int e = 0; e < EP; ++e) o[e] = Elem<T>::from_f(Elem<T>::to_f(hv[e]) + Elem<T>::to_f(sv[e]));
  }
  return o;
}
template <typename T, int SRC>
__global__ __launch_bounds__(256) void norm_partial_src_kernel(const T* __restrict__ xa, const T* __restrict__ xh, T* __restrict__ y,
                                                               float* __restrict__ partial, int H, int W, int C, int nchunks) {
  constexpr int EP = Elem<T>::kPer16B;
  typedef typename Vec16<T>::type V;
  extern __shared__ float red[];  // [pl][2][C]
  const int HW = H * W;
  const int chunk = blockIdx.x, b = blockIdx.y;
  const int cpp = C / EP, pl = 256 / cpp;
  const int cq = threadIdx.x % cpp, pj = threadIdx.x / cpp;
  const int per = (HW + nchunks - 1) / nchunks;
  const int p0 = chunk * per, p1 = min(HW, p0 + per);
  T* base = y + (size_t)b * HW * C;
  float s1[EP], s2[EP], k[EP];
#pragma unroll
  for (int e = 0; e < EP; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  if (pj < pl) {
    const V kv = norm_src_value<T, SRC>(xa, xh, b, 0, H, W, C, cq);      // shift: pixel 0 of the sample (computed, not read back)
#pragma unroll
    for (int e = 0; e < EP; ++e) k[e] = Elem<T>::to_f(kv[e]);
#pragma unroll 4
    for (int pp = p0 + pj; pp < p1; pp += pl) {
      const V v = norm_src_value<T, SRC>(xa, xh, b, pp, H, W, C, cq);
      *reinterpret_cast<V*>(base + (size_t)pp * C + cq * EP) = v;
#pragma unroll
      for (int e = 0; e < EP; ++e) {
        const float d = Elem<T>::to_f(v[e]) - k[e];
        s1[e] += d;
        s2[e] = fmaf(d, d, s2[e]);
      }
    }
#pragma unroll
    for (int e = 0; e < EP; ++e) {
      red[(pj * 2 + 0) * C + cq * EP + e] = s1[e];
      red[(pj * 2 + 1) * C + cq * EP + e] = s2[e];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    float s = 0.f;
    for (int j = 0; j < pl; ++j) s += red[j * 2 * C + i];
    partial[((size_t)(b * nchunks + chunk) * 2) * C + i] = s;
  }
}

// One thread per (b,c) [instance] or per c [batch].  Writes mean/rstd/scale/shift [B,C].
template <typename T>
__global__ void norm_finalize_kernel(const T* __restrict__ y, const float* __restrict__ partial,
                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                     float* __restrict__ state, float* __restrict__ running_mean,
                                     float* __restrict__ running_var, int B, int HW, int C, int nchunks, int batch_mode,
                                     float eps, float momentum) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  float* __restrict__ mean_o = state;                       // state: [4][B][C] = mean, rstd, scale, beta
  float* __restrict__ rstd_o = state + (size_t)B * C;
  float* __restrict__ scale_o = state + (size_t)2 * B * C;
  float* __restrict__ shift_o = state + (size_t)3 * B * C;
  if (!batch_mode) {
    if (idx >= B * C) return;
    const int b = idx / C, c = idx - b * C;
    // fixed summation order, but 8 loads in flight per accumulator chain (a plain loop waits for every load in turn)
    float s1 = 0.f, s2 = 0.f;
    const float* pp = partial + ((size_t)b * nchunks * 2) * C + c;
    int k = 0;
    for (; k + 8 <= nchunks; k += 8) {
      float a[8], q[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { a[u] = pp[((size_t)(k + u) * 2 + 0) * C]; q[u] = pp[((size_t)(k + u) * 2 + 1) * C]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) { s1 += a[u]; s2 += q[u]; }
    }
    for (; k < nchunks; ++k) { s1 += pp[((size_t)k * 2 + 0) * C]; s2 += pp[((size_t)k * 2 + 1) * C]; }
    const float n = (float)HW;
    const float kk = Elem<T>::to_f(y[(size_t)b * HW * C + c]);
    const float m1 = s1 / n;
    const float var = fmaxf(s2 / n - m1 * m1, 0.f);
    const float mean = kk + m1, rstd = 1.f / sqrtf(var + eps);
    const float sc = gamma[c] * rstd;
    mean_o[idx] = mean; rstd_o[idx] = rstd; scale_o[idx] = sc; shift_o[idx] = beta[c];
  } else {
    if (idx >= C) return;
    const int c = idx;
    float s1 = 0.f, s2 = 0.f;
    for (int k = 0; k < B * nchunks; ++k) {
      s1 += partial[((size_t)k * 2 + 0) * C + c];
      s2 += partial[((size_t)k * 2 + 1) * C + c];
    }
    const float n = (float)B * (float)HW;
    const float kk = Elem<T>::to_f(y[c]);
    const float m1 = s1 / n;
    const float var = fmaxf(s2 / n - m1 * m1, 0.f);
    const float mean = kk + m1, rstd = 1.f / sqrtf(var + eps);
    const float sc = gamma[c] * rstd, sh = beta[c];
    for (int b = 0; b < B; ++b) {
      mean_o[b * C + c] = mean; rstd_o[b * C + c] = rstd; scale_o[b * C + c] = sc; shift_o[b * C + c] = sh;
    }
    if (running_mean) {  // torch.nn.BatchNorm2d: unbiased variance in the running estimate
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * (n / fmaxf(n - 1.f, 1.f));
    }
  }
}

// InstanceNorm statistics in ONE launch: every block writes the partial sums of its pixel chunk, then the block that
// arrives last for sample b (ticket from an agent-scope counter) combines the chunks of b in a fixed order and writes the
// [4][B][C] state.  Hand-off protocol (cdna_hip_programming.md, Guideline 16): plain stores -> every wave s_waitcnt vmcnt(0)
// -> workgroup barrier -> lane 0: agent release fence, vmcnt(0), relaxed agent fetch_add; the last arriver: agent acquire
// fence, vmcnt(0), barrier, plain loads.  The counter is reset by the last arriver (it must be zero before the first use).
template <typename T, bool SINGLE>
__global__ __launch_bounds__(256) void norm_stats_fused_kernel(const T* __restrict__ y, float* __restrict__ partial,
                                                               int* __restrict__ counters, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float* __restrict__ state, int B,
                                                               int HW, int C, int nchunks, float eps) {
  constexpr int EP = Elem<T>::kPer16B;
  typedef typename Vec16<T>::type V;
  extern __shared__ float red[];  // [pl][2][C]
  __shared__ int s_last;
  const int chunk = blockIdx.x, b = blockIdx.y;
  const int cpp = C / EP, pl = 256 / cpp;
  const int cq = threadIdx.x % cpp, pj = threadIdx.x / cpp;
  const int per = (HW + nchunks - 1) / nchunks;
  const int p0 = chunk * per, p1 = min(HW, p0 + per);
  const T* base = y + (size_t)b * HW * C;
  float s1[EP], s2[EP], k[EP];
#pragma unroll
  for (int e = 0; e < EP; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  if (pj < pl) {
    V kv = *reinterpret_cast<const V*>(base + cq * EP);
#pragma unroll
    for (int e = 0; e < EP; ++e) k[e] = Elem<T>::to_f(kv[e]);
#pragma unroll 4
    for (int pp = p0 + pj; pp < p1; pp += pl) {
      V v = *reinterpret_cast<const V*>(base + (size_t)pp * C + cq * EP);
#pragma unroll
      for (int e = 0; e < EP; ++e) {
        const float d = Elem<T>::to_f(v[e]) - k[e];
        s1[e] += d;
        s2[e] = fmaf(d, d, s2[e]);
      }
    }
#pragma unroll
    for (int e = 0; e < EP; ++e) {
      red[(pj * 2 + 0) * C + cq * EP + e] = s1[e];
      red[(pj * 2 + 1) * C + cq * EP + e] = s2[e];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    float s = 0.f;
    for (int j = 0; j < pl; ++j) s += red[j * 2 * C + i];
    // (hand-off form: write-through `sc1` stores -- the bytes are in memory when the store is acknowledged, no L2 write-back fence per block)
    if constexpr (SINGLE) partial[((size_t)(b * nchunks + chunk) * 2) * C + i] = s;
    else __hip_atomic_store(&partial[((size_t)(b * nchunks + chunk) * 2) * C + i], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if constexpr (SINGLE) {   // one block owns the whole sample: its own partials are visible after a workgroup barrier
    __syncthreads();
    const size_t plane1 = (size_t)B * C;
    for (int c = threadIdx.x; c < C; c += 256) {
      const float a1 = partial[((size_t)b * 2 + 0) * C + c], a2 = partial[((size_t)b * 2 + 1) * C + c];
      const float n = (float)HW;
      const float k0 = Elem<T>::to_f(base[c]);
      const float m1 = a1 / n;
      const float var = fmaxf(a2 / n - m1 * m1, 0.f);
      const float mean = k0 + m1, rstd = 1.f / sqrtf(var + eps);
      const int idx = b * C + c;
      state[idx] = mean; state[plane1 + idx] = rstd; state[2 * plane1 + idx] = gamma[c] * rstd; state[3 * plane1 + idx] = beta[c];
    }
    return;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    // (round 6: no release fence -- every partial was stored `sc1` and drained by its wave's vmcnt(0) above, behind the barrier;
    // MI355X_MICROARCH.md, Valid forms.  The LAST arriver still runs the acquire: several of these blocks share a CU.)
    const int t = __hip_atomic_fetch_add(&counters[b], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = t == nchunks - 1;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(&counters[b], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    s_last = last;
  }
  __syncthreads();
  if (!s_last) return;
  const size_t plane = (size_t)B * C;
  for (int c = threadIdx.x; c < C; c += 256) {
    float a1 = 0.f, a2 = 0.f;
    int kk = 0;
    for (; kk + 8 <= nchunks; kk += 8) {      // (8 loads of a sum in flight, added in norm_finalize_kernel's order; `sc1` loads: L2-served)
      float u1[8], u2[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        u1[u] = __hip_atomic_load(&partial[((size_t)(b * nchunks + kk + u) * 2 + 0) * C + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        u2[u] = __hip_atomic_load(&partial[((size_t)(b * nchunks + kk + u) * 2 + 1) * C + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { a1 += u1[u]; a2 += u2[u]; }
    }
    for (; kk < nchunks; ++kk) {
      a1 += __hip_atomic_load(&partial[((size_t)(b * nchunks + kk) * 2 + 0) * C + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      a2 += __hip_atomic_load(&partial[((size_t)(b * nchunks + kk) * 2 + 1) * C + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const float n = (float)HW;
    const float k0 = Elem<T>::to_f(base[c]);
    const float m1 = a1 / n;
    const float var = fmaxf(a2 / n - m1 * m1, 0.f);
    const float mean = k0 + m1, rstd = 1.f / sqrtf(var + eps);
    const int idx = b * C + c;
    state[idx] = mean; state[plane + idx] = rstd; state[2 * plane + idx] = gamma[c] * rstd; state[3 * plane + idx] = beta[c];
  }
}

// eval-mode BatchNorm: scale/shift from the running statistics
__global__ void norm_eval_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                 const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                 float* __restrict__ state, int B, int C, float eps) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * C) return;
  const int c = idx % C;
  const float rstd = 1.f / sqrtf(running_var[c] + eps), sc = gamma[c] * rstd;
  state[idx] = running_mean[c]; state[(size_t)B * C + idx] = rstd; state[(size_t)2 * B * C + idx] = sc;
  state[(size_t)3 * B * C + idx] = beta[c];
}

// ---- backward ----
// g = dL/d relu(norm(y));  gm = g * [norm(y) > 0];  xn = (y - mean)*rstd
// partial sums per chunk: s1 = sum gm, s2 = sum gm*xn
template <typename T>
__global__ __launch_bounds__(256) void norm_bwd_partial_kernel(const T* __restrict__ g, const T* __restrict__ y,
                                                               const float* __restrict__ state, int B,
                                                               float* __restrict__ partial, int HW, int C, int nchunks,
                                                               int relu) {
  const float* __restrict__ mean = state;
  const float* __restrict__ rstd = state + (size_t)B * C;
  const float* __restrict__ scale = state + (size_t)2 * B * C;
  const float* __restrict__ shift = state + (size_t)3 * B * C;
  constexpr int EP = Elem<T>::kPer16B;
  typedef typename Vec16<T>::type V;
  extern __shared__ float red[];
  const int chunk = blockIdx.x, b = blockIdx.y;
  const int cpp = C / EP, pl = 256 / cpp;
  const int cq = threadIdx.x % cpp, pj = threadIdx.x / cpp;
  const int per = (HW + nchunks - 1) / nchunks;
  const int p0 = chunk * per, p1 = min(HW, p0 + per);
  const size_t base = (size_t)b * HW * C;
  float s1[EP], s2[EP];
#pragma unroll
  for (int e = 0; e < EP; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  if (pj < pl) {
    float mu[EP], rs[EP], sc[EP], sh[EP];
#pragma unroll
    for (int e = 0; e < EP; ++e) {
      const int c = b * C + cq * EP + e;
      mu[e] = mean[c]; rs[e] = rstd[c]; sc[e] = scale[c]; sh[e] = shift[c];
    }
#pragma unroll 4
    for (int pp = p0 + pj; pp < p1; pp += pl) {
      V gv = *reinterpret_cast<const V*>(g + base + (size_t)pp * C + cq * EP);
      V yv = *reinterpret_cast<const V*>(y + base + (size_t)pp * C + cq * EP);
#pragma unroll
      for (int e = 0; e < EP; ++e) {
        const float yy = Elem<T>::to_f(yv[e]);
        float gg = Elem<T>::to_f(gv[e]);
        if (relu && !(fmaf(yy - mu[e], sc[e], sh[e]) > 0.f)) gg = 0.f;
        s1[e] += gg;
        s2[e] = fmaf(gg, (yy - mu[e]) * rs[e], s2[e]);
      }
    }
#pragma unroll
    for (int e = 0; e < EP; ++e) {
      red[(pj * 2 + 0) * C + cq * EP + e] = s1[e];
      red[(pj * 2 + 1) * C + cq * EP + e] = s2[e];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    float s = 0.f;
    for (int j = 0; j < pl; ++j) s += red[j * 2 * C + i];
    partial[((size_t)(b * nchunks + chunk) * 2) * C + i] = s;
  }
}

// One block per 8 channels: thread (b-lane, c) sums the pixel-chunk partials of its (b, c) in a fixed order -> S1,S2[b,c]
// (divided by HW for instance norm), then the block reduces over b for dgamma / dbeta (and, for batch norm, overwrites
// S1,S2 with the batch-wide means).  blockDim = 256 = 32 b-lanes x 8 channels.
__device__ __forceinline__ void norm_bwd_sum_body(const float* __restrict__ partial, float* __restrict__ S1,
                                                  float* __restrict__ S2, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                  int B, int HW, int C, int nchunks, int batch_mode, int accumulate, int bx) {
  __shared__ float r1[32][9], r2[32][9];
  const int cl = threadIdx.x & 7, bl = threadIdx.x >> 3;
  const int c = bx * 8 + cl;
  float t1 = 0.f, t2 = 0.f;
  if (c < C) {
    for (int b = bl; b < B; b += 32) {
      float s1 = 0.f, s2 = 0.f;
      const float* pp = partial + ((size_t)b * nchunks * 2) * C + c;
      int k = 0;
      // 32 chunks' loads in flight at a time (the chain kernel this is: 16 blocks, one L2 round trip per batch), summed in the
      // same order as before
      for (; k + 32 <= nchunks; k += 32) {
        float a[32], q[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) { a[u] = pp[((size_t)(k + u) * 2 + 0) * C]; q[u] = pp[((size_t)(k + u) * 2 + 1) * C]; }
#pragma unroll
        for (int u = 0; u < 32; ++u) { s1 += a[u]; s2 += q[u]; }
      }
      for (; k + 8 <= nchunks; k += 8) {
        float a[8], q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { a[u] = pp[((size_t)(k + u) * 2 + 0) * C]; q[u] = pp[((size_t)(k + u) * 2 + 1) * C]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) { s1 += a[u]; s2 += q[u]; }
      }
      for (; k < nchunks; ++k) { s1 += pp[((size_t)k * 2 + 0) * C]; s2 += pp[((size_t)k * 2 + 1) * C]; }
      t1 += s1; t2 += s2;
      if (!batch_mode && S1) { S1[b * C + c] = s1 / (float)HW; S2[b * C + c] = s2 / (float)HW; }
    }
  }
  r1[bl][cl] = t1; r2[bl][cl] = t2;
  __syncthreads();
  if (bl == 0 && c < C) {
    float a1 = 0.f, a2 = 0.f;
    for (int j = 0; j < 32; ++j) { a1 += r1[j][cl]; a2 += r2[j][cl]; }
    dgamma[c] = accumulate ? dgamma[c] + a2 : a2;
    dbeta[c] = accumulate ? dbeta[c] + a1 : a1;
    r1[0][cl] = a1; r2[0][cl] = a2;
  }
  __syncthreads();
  if (batch_mode && c < C) {
    const float n = (float)B * (float)HW;
    for (int b = bl; b < B; b += 32) { S1[b * C + c] = r1[0][cl] / n; S2[b * C + c] = r2[0][cl] / n; }
  }
}

__global__ __launch_bounds__(256) void norm_bwd_sum_kernel(const float* __restrict__ partial, float* __restrict__ S1,
                                                           float* __restrict__ S2, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           int B, int HW, int C, int nchunks, int batch_mode, int accumulate) {
  norm_bwd_sum_body(partial, S1, S2, dgamma, dbeta, B, HW, C, nchunks, batch_mode, accumulate, blockIdx.x);
}
// Two norm backwards of ONE shape in one launch each (blockIdx.y / .z picks the job): the two regression heads of a stage are
// back-propagated in lock-step (model.py:54-65 / :103-114), and on the heads' stretch of the backward pass the chain runs alone -- every
// launch boundary there is exposed (round 4).  Same bodies, same bits as two single calls.
struct NormBwdJob { const void* g; const void* y; const float* state; const float* partial; float* S1; float* S2; void* dy; float* dgamma; float* dbeta; };
struct NormBwdPair { NormBwdJob j[2]; };
__global__ __launch_bounds__(256) void norm_bwd_sum_pair_kernel(NormBwdPair q, int B, int HW, int C, int nchunks, int accumulate) {
  const NormBwdJob& j = q.j[blockIdx.y];
  norm_bwd_sum_body(j.partial, j.S1, j.S2, j.dgamma, j.dbeta, B, HW, C, nchunks, 0, accumulate, blockIdx.x);
}

// dy = gamma*rstd * (gm - S1 - xn*S2) (+ addend).  Same (chunk, b) decomposition as the partial kernels so that the
// per-channel constants are loaded once per thread, not once per element.
// FROM_PARTIAL (instance norm): every block sums the pixel-chunk partials of its sample itself (fixed order, so all
// blocks of a sample get identical sums) instead of waiting for a separate reduction launch.
template <typename T, bool FROM_PARTIAL>
__device__ __forceinline__ void norm_bwd_apply_body(const T* __restrict__ g, const T* __restrict__ y,
                                                    const float* __restrict__ state, int B,
                                                    const float* __restrict__ S1, const float* __restrict__ S2,
                                                    const T* __restrict__ addend, T* __restrict__ dy, int HW, int C,
                                                    int nchunks, int relu, int chunk, int b) {
  constexpr int EP = Elem<T>::kPer16B;
  typedef typename Vec16<T>::type V;
  const int cpp = C / EP, pl = 256 / cpp;
  const int cq = threadIdx.x % cpp, pj = threadIdx.x / cpp;
  if (pj >= pl) return;
  const int per = (HW + nchunks - 1) / nchunks;
  const int p0 = chunk * per, p1 = min(HW, p0 + per);
  const size_t base = (size_t)b * HW * C;
  const size_t plane = (size_t)B * C;
  float mu[EP], rs[EP], sc[EP], sh[EP], s1[EP], s2[EP];
#pragma unroll
  for (int e = 0; e < EP; ++e) {
    const int c = b * C + cq * EP + e;
    mu[e] = state[c]; rs[e] = state[plane + c]; sc[e] = state[2 * plane + c]; sh[e] = state[3 * plane + c];
    if constexpr (!FROM_PARTIAL) { s1[e] = S1[c]; s2[e] = S2[c]; }
  }
  if constexpr (FROM_PARTIAL) {   // S1 = the partial slab [B][nchunks][2][C]
#pragma unroll
    for (int e = 0; e < EP; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    const float* pp0 = S1 + ((size_t)b * nchunks * 2) * C + cq * EP;
#pragma unroll 4
    for (int k = 0; k < nchunks; ++k) {
#pragma unroll
      for (int e = 0; e < EP; e += 4) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(pp0 + ((size_t)k * 2 + 0) * C + e);
        const f32x4 q = *reinterpret_cast<const f32x4*>(pp0 + ((size_t)k * 2 + 1) * C + e);
        s1[e] += a.x; s1[e + 1] += a.y; s1[e + 2] += a.z; s1[e + 3] += a.w;
        s2[e] += q.x; s2[e + 1] += q.y; s2[e + 2] += q.z; s2[e + 3] += q.w;
      }
    }
    const float inv = 1.f / (float)HW;
#pragma unroll
    for (int e = 0; e < EP; ++e) { s1[e] *= inv; s2[e] *= inv; }
  }
#pragma unroll 4
  for (int pp = p0 + pj; pp < p1; pp += pl) {
    const size_t off = base + (size_t)pp * C + cq * EP;
    V gv = *reinterpret_cast<const V*>(g + off);
    V yv = *reinterpret_cast<const V*>(y + off);
    V av = {};
    if (addend) av = *reinterpret_cast<const V*>(addend + off);
    V o;
#pragma unroll
    for (int e = 0; e < EP; ++e) {
      const float yy = Elem<T>::to_f(yv[e]);
      float gg = Elem<T>::to_f(gv[e]);
      if (relu && !(fmaf(yy - mu[e], sc[e], sh[e]) > 0.f)) gg = 0.f;
      const float xn = (yy - mu[e]) * rs[e];
      float r = sc[e] * (gg - s1[e] - xn * s2[e]);     // scale = gamma*rstd
      if (addend) r += Elem<T>::to_f(av[e]);
      o[e] = Elem<T>::from_f(r);
    }
    *reinterpret_cast<V*>(dy + off) = o;
  }
}

template <typename T, bool FROM_PARTIAL>
__global__ __launch_bounds__(256) void norm_bwd_apply_kernel(const T* __restrict__ g, const T* __restrict__ y,
                                                             const float* __restrict__ state, int B,
                                                             const float* __restrict__ S1, const float* __restrict__ S2,
                                                             const T* __restrict__ addend, T* __restrict__ dy, int HW, int C,
                                                             int nchunks, int relu) {
  norm_bwd_apply_body<T, FROM_PARTIAL>(g, y, state, B, S1, S2, addend, dy, HW, C, nchunks, relu, blockIdx.x, blockIdx.y);
}
template <typename T>
__global__ __launch_bounds__(256) void norm_bwd_apply_pair_kernel(NormBwdPair q, int B, int HW, int C, int nchunks, int relu) {
  const NormBwdJob& j = q.j[blockIdx.z];
  norm_bwd_apply_body<T, false>((const T*)j.g, (const T*)j.y, j.state, B, j.S1, j.S2, (const T*)nullptr, (T*)j.dy, HW, C, nchunks, relu, blockIdx.x,
                                blockIdx.y);
}

// Round 6 -- the reduction launch taken off the chain (instance norm).  norm_bwd_sum_kernel is a 16-block launch of ~8 us whose only
// consumer on the critical path is the apply launch behind it (S1, S2); its other outputs -- dgamma, dbeta -- feed the flat gradient only.
// Here every workgroup of the apply launch sums the `pchunks` slab rows of ITS sample itself: 2 C sums, one per thread, every load of a sum
// issued before the first add (32 in flight: ONE L2 round trip instead of 32 dependent ones -- round 5's experiment 13 walked them one
// after the other in every one of the 1024 workgroups and lost 3.5 % of the step), added in norm_bwd_sum_body's order (k ascending), divided
// by HW like there: S1, S2 and therefore dy are the same bits.  The parameter sums run on a side stream (norm_bwd_sum_kernel with
// S1 = null) from the same slab, which therefore has to be the layer's own (the engine allocates one per norm).
template <typename T, int NTL = 0>
__device__ __forceinline__ void norm_bwd_apply_fold_body(const T* __restrict__ g, const T* __restrict__ y, const float* __restrict__ state, int B,
                                                         const float* __restrict__ partial, int pchunks, const T* __restrict__ addend,
                                                         T* __restrict__ dy, int HW, int C, int nchunks, int relu, int chunk, int b) {
  constexpr int EP = Elem<T>::kPer16B;
  typedef typename Vec16<T>::type V;
  __shared__ float ssum[1024];     // [2][C], C <= 512
  // the per-channel state of this thread's slot: requested BEFORE the slab sums, so that its round trip and the sums' overlap (behind the
  // barrier below they were one more dependent L2 round trip in every workgroup)
  const int cpp = C / EP, pl = 256 / cpp;
  const int cq = threadIdx.x % cpp, pj = threadIdx.x / cpp;
  const size_t plane = (size_t)B * C;
  float mu[EP], rs[EP], sc[EP], sh[EP];
  if (pj < pl) {
#pragma unroll
    for (int e = 0; e < EP; ++e) {
      const int c = b * C + cq * EP + e;
      mu[e] = state[c]; rs[e] = state[plane + c]; sc[e] = state[2 * plane + c]; sh[e] = state[3 * plane + c];
    }
  }
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    const int qq = i >= C ? 1 : 0, c = i - qq * C;
    const float* pp = partial + ((size_t)b * pchunks * 2 + qq) * C + c;      // slab row k of this sum: pp[k * 2 C]
    const size_t st = (size_t)2 * C;
    float sacc = 0.f;
    int k = 0;
    for (; k + 32 <= pchunks; k += 32) {
      float a[32];
#pragma unroll
      for (int u = 0; u < 32; ++u) a[u] = pp[(size_t)(k + u) * st];
#pragma unroll
      for (int u = 0; u < 32; ++u) sacc += a[u];
    }
    for (; k + 8 <= pchunks; k += 8) {
      float a[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] = pp[(size_t)(k + u) * st];
#pragma unroll
      for (int u = 0; u < 8; ++u) sacc += a[u];
    }
    for (; k < pchunks; ++k) sacc += pp[(size_t)k * st];
    ssum[i] = sacc / (float)HW;
  }
  __syncthreads();
  if (pj >= pl) return;
  const int per = (HW + nchunks - 1) / nchunks;
  const int p0 = chunk * per, p1 = min(HW, p0 + per);
  const size_t base = (size_t)b * HW * C;
  float s1[EP], s2[EP];
#pragma unroll
  for (int e = 0; e < EP; ++e) { s1[e] = ssum[cq * EP + e]; s2[e] = ssum[C + cq * EP + e]; }
#pragma unroll 4
  for (int pp = p0 + pj; pp < p1; pp += pl) {
    const size_t off = base + (size_t)pp * C + cq * EP;
    // (NTL bits, debug build's PWR_FOLD_NT: 1 = g and y read non-temporally -- each is read once here --, 2 = dy stored non-temporally)
    V gv = (NTL & 1) ? __builtin_nontemporal_load(reinterpret_cast<const V*>(g + off)) : *reinterpret_cast<const V*>(g + off);
    V yv = (NTL & 1) ? __builtin_nontemporal_load(reinterpret_cast<const V*>(y + off)) : *reinterpret_cast<const V*>(y + off);
    V av = {};
    if (addend) av = *reinterpret_cast<const V*>(addend + off);
    V o;
#pragma unroll
    for (int e = 0; e < EP; ++e) {
      const float yy = Elem<T>::to_f(yv[e]);
      float gg = Elem<T>::to_f(gv[e]);
      if (relu && !(fmaf(yy - mu[e], sc[e], sh[e]) > 0.f)) gg = 0.f;
      const float xn = (yy - mu[e]) * rs[e];
      float r = sc[e] * (gg - s1[e] - xn * s2[e]);     // scale = gamma*rstd  (norm_bwd_apply_body's expression)
      if (addend) r += Elem<T>::to_f(av[e]);
      o[e] = Elem<T>::from_f(r);
    }
    if (NTL & 2) __builtin_nontemporal_store(o, reinterpret_cast<V*>(dy + off));
    else *reinterpret_cast<V*>(dy + off) = o;
  }
}
struct NormBwdFoldJob { const void* g; const void* y; const float* state; const float* partial; const void* addend; void* dy; };
struct NormBwdFoldPair { NormBwdFoldJob j[2]; };
template <typename T, int NTL = 0>
__global__ __launch_bounds__(256) void norm_bwd_apply_fold_kernel(NormBwdFoldPair q, int B, int HW, int C, int nchunks, int pchunks, int relu) {
  const NormBwdFoldJob& j = q.j[blockIdx.z];
  norm_bwd_apply_fold_body<T, NTL>((const T*)j.g, (const T*)j.y, j.state, B, j.partial, pchunks, (const T*)j.addend, (T*)j.dy, HW, C, nchunks, relu,
                              blockIdx.x, blockIdx.y);
}

// The parameter sums of SEVERAL norm backwards in one launch (round 6): as a launch of its own per layer -- 16 blocks, ~8 us -- the folded
// form's parameter sums added 28 small kernels per step to the side streams and the step got LONGER than with the reduction on the chain
// (5.08 against 5.02 ms: the step ends when the side streams do).  Block x looks its job up in a prefix table and runs norm_bwd_sum_body
// without the per-sample outputs: the same order, the same bits as pwr_norm_bwd_from_partial's dgamma / dbeta.
struct NormParamGroup {
  static constexpr int kMax = 40;
  const float* partial[kMax]; float* dgamma[kMax]; float* dbeta[kMax];
  int HW[kMax], C[kMax], chunks[kMax], first[kMax + 1];
  int n, B, accumulate;
};
__global__ __launch_bounds__(256) void norm_bwd_params_group_kernel(NormParamGroup g) {
  int j = 0;
  while (j + 1 < g.n && (int)blockIdx.x >= g.first[j + 1]) ++j;
  norm_bwd_sum_body(g.partial[j], nullptr, nullptr, g.dgamma[j], g.dbeta[j], g.B, g.HW[j], g.C[j], g.chunks[j], 0, g.accumulate, blockIdx.x - g.first[j]);
}

// out = bf16 / fp32 of relu(norm(y)) as the convs and weight gradients compute it on operand load -- fmaf(y - mean, scale, beta), ReLU,
// ONE rounding -- written out as a tensor (round 6).  The wave-specialised weight gradient of the heads' norm-fed layers spends a third of
// its loader's issue slots on exactly this arithmetic, three times over (once per kernel row) and once per split: 124.5 us in the step
// against 84.9 us for the same layer without a norm.  Materialising the operand once on the weight gradient's own (side) stream -- 67 MB
// of traffic, ~15 us -- and running the plain form is the same LDS tile, the same slabs, the same dW bit for bit.
template <typename T>
__global__ __launch_bounds__(256) void norm_apply_kernel(const T* __restrict__ y, const float* __restrict__ state, T* __restrict__ out, int B,
                                                         int HW, int C, int nchunks, int relu) {
  constexpr int EP = Elem<T>::kPer16B;
  typedef typename Vec16<T>::type V;
  const int chunk = blockIdx.x, b = blockIdx.y;
  const int cpp = C / EP, pl = 256 / cpp;
  const int cq = threadIdx.x % cpp, pj = threadIdx.x / cpp;
  if (pj >= pl) return;
  const int per = (HW + nchunks - 1) / nchunks;
  const int p0 = chunk * per, p1 = min(HW, p0 + per);
  const size_t base = (size_t)b * HW * C, plane = (size_t)B * C;
  float mu[EP], sc[EP], be[EP];
#pragma unroll
  for (int e = 0; e < EP; ++e) {
    const int c = b * C + cq * EP + e;
    mu[e] = state[c]; sc[e] = state[2 * plane + c]; be[e] = state[3 * plane + c];
  }
#pragma unroll 4
  for (int pp = p0 + pj; pp < p1; pp += pl) {
    const size_t off = base + (size_t)pp * C + cq * EP;
    const V v = __builtin_nontemporal_load(reinterpret_cast<const V*>(y + off));
    V o;
#pragma unroll
    for (int e = 0; e < EP; ++e) {
      float f = fmaf(Elem<T>::to_f(v[e]) - mu[e], sc[e], be[e]);
      if (relu) f = fmaxf(f, 0.f);
      o[e] = Elem<T>::from_f(f);
    }
    *reinterpret_cast<V*>(out + off) = o;
  }
}

static inline int norm_chunks(int B, int HW) {
  int n = (1024 + B - 1) / B;
  int maxc = (HW + 31) / 32;
  if (n > maxc) n = maxc;
  if (n < 1) n = 1;
  if (n > 64) n = 64;
  return n;
}

}  // namespace pwr

using namespace pwr;

extern "C" int pwr_norm_chunks(int B, int HW) { return norm_chunks(B, HW); }

// workspace = [counters: B + 1 ints, padded to 256 B, must be zero before the first use (they reset themselves)] [partials]
static inline size_t norm_counter_bytes(int B) { return ((size_t)(B + 1) * 4 + 255) / 256 * 256; }
extern "C" size_t pwr_norm_partial_bytes(int B, int HW, int C) {
  return norm_counter_bytes(B) + (size_t)B * norm_chunks(B, HW) * 2 * C * sizeof(float);
}

// mode: 0 instance, 1 batch (training statistics), 2 batch eval (running statistics)
extern "C" int pwr_norm_stats(const void* y, const float* gamma, const float* beta, float* running_mean, float* running_var,
                              float* partial, float* state, int B, int HW, int C, int mode, float eps, float momentum,
                              int dtype, void* stream) {
  const int EP = dtype == PWR_BF16 ? 8 : 4;
  if (C % EP || C / EP > 256) return PWR_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (mode == 2) {
    hipLaunchKernelGGL(norm_eval_kernel, dim3((B * C + 255) / 256), dim3(256), 0, s, gamma, beta, running_mean, running_var,
                       state, B, C, eps);
    return (int)hipGetLastError();
  }
  const int nch = norm_chunks(B, HW);
  const int pl = 256 / (C / EP);
  const size_t sh = (size_t)pl * 2 * C * 4;
  const int n = mode == 1 ? C : B * C;
  int* counters = reinterpret_cast<int*>(partial);
  partial = reinterpret_cast<float*>(reinterpret_cast<char*>(partial) + norm_counter_bytes(B));
  // Measured on MI355X (BASELINE C2): the in-kernel hand-off costs more than the launch it saves (1024 blocks x release
  // fence): train step 13.0 ms fused vs 12.6 ms as two launches.  So it is opt-in; small maps, where ONE block owns the
  // whole sample and no hand-off is needed, always take the single-launch form.
  // (round 6: the hand-off form without the release fence -- `sc1` partial stores -- behind PWR_NORM_FUSE, debug build; measured below)
  static const bool fuse = PWR_DBG_ENV("PWR_NORM_FUSE", 0) != 0;
  static const int fwd_small = PWR_DBG_ENV("PWR_NORM_FWD_SMALL", 512);
  if (mode == 0 && HW <= fwd_small) {
    if (dtype == PWR_BF16) hipLaunchKernelGGL((norm_stats_fused_kernel<bf16_t, true>), dim3(1, B), dim3(256), sh, s, (const bf16_t*)y, partial, counters, gamma, beta, state, B, HW, C, 1, eps);
    else hipLaunchKernelGGL((norm_stats_fused_kernel<float, true>), dim3(1, B), dim3(256), sh, s, (const float*)y, partial, counters, gamma, beta, state, B, HW, C, 1, eps);
    return (int)hipGetLastError();
  }
  if (mode == 0 && fuse) {
    if (dtype == PWR_BF16) hipLaunchKernelGGL((norm_stats_fused_kernel<bf16_t, false>), dim3(nch, B), dim3(256), sh, s, (const bf16_t*)y, partial, counters, gamma, beta, state, B, HW, C, nch, eps);
    else hipLaunchKernelGGL((norm_stats_fused_kernel<float, false>), dim3(nch, B), dim3(256), sh, s, (const float*)y, partial, counters, gamma, beta, state, B, HW, C, nch, eps);
    return (int)hipGetLastError();
  }
  if (dtype == PWR_BF16) {
    hipLaunchKernelGGL((norm_partial_kernel<bf16_t>), dim3(nch, B), dim3(256), sh, s, (const bf16_t*)y, partial, HW, C, nch, mode);
    hipLaunchKernelGGL((norm_finalize_kernel<bf16_t>), dim3((n + 255) / 256), dim3(256), 0, s, (const bf16_t*)y, partial, gamma,
                       beta, state, running_mean, running_var, B, HW, C, nch, mode, eps, momentum);
  } else {
    hipLaunchKernelGGL((norm_partial_kernel<float>), dim3(nch, B), dim3(256), sh, s, (const float*)y, partial, HW, C, nch, mode);
    hipLaunchKernelGGL((norm_finalize_kernel<float>), dim3((n + 255) / 256), dim3(256), 0, s, (const float*)y, partial, gamma,
                       beta, state, running_mean, running_var, B, HW, C, nch, mode, eps, momentum);
  }
  return (int)hipGetLastError();
}

namespace pwr {
// Statistics from the per-tile sums that a conv epilogue wrote (pwr_conv_fwd_stats): `chunks` entries per sample, each
// (sum (v - k), sum (v - k)^2, k) over HW/chunks pixels with its own shift k.  Chunks are merged like Chan et al.'s
// parallel variance (mean and M2 per chunk) -- no cancellation anywhere.
__global__ void norm_finalize_chunks_kernel(const float* __restrict__ partial, const float* __restrict__ gamma, const float* __restrict__ beta,
                                            float* __restrict__ state, float* __restrict__ running_mean, float* __restrict__ running_var,
                                            int B, int HW, int C, int chunks, int batch_mode, float eps, float momentum) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (batch_mode ? C : B * C)) return;
  const int b = batch_mode ? 0 : idx / C, c = batch_mode ? idx : idx - b * C;
  const int total = batch_mode ? B * chunks : chunks;
  const float* base = partial + ((size_t)b * chunks * 3) * C + c;
  const float nper = (float)(HW / chunks), inv = 1.f / nper;
  // one pass, chunks merged one after the other (equal sizes): delta = chunk mean - running mean,
  // mean += delta / (j+1), M2 += M2_j + n * delta^2 * j / (j+1); 8 chunks' loads in flight at a time
  float mean = 0.f, m2 = 0.f, cnt = 0.f;
  auto merge = [&](float a, float q, float k) {
    const float mj = k + a * inv, d = mj - mean, c1 = cnt + 1.f;
    mean += d / c1;
    m2 += fmaxf(q - a * a * inv, 0.f) + nper * d * d * (cnt / c1);
    cnt = c1;
  };
  int j = 0;
  for (; j + 32 <= total; j += 32) {       // 32 chunks' loads in flight at a time, merged in the same order as before
    float a[32], q[32], k[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      a[u] = base[((size_t)(j + u) * 3 + 0) * C]; q[u] = base[((size_t)(j + u) * 3 + 1) * C]; k[u] = base[((size_t)(j + u) * 3 + 2) * C];
    }
#pragma unroll
    for (int u = 0; u < 32; ++u) merge(a[u], q[u], k[u]);
  }
  for (; j + 8 <= total; j += 8) {
    float a[8], q[8], k[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a[u] = base[((size_t)(j + u) * 3 + 0) * C]; q[u] = base[((size_t)(j + u) * 3 + 1) * C]; k[u] = base[((size_t)(j + u) * 3 + 2) * C];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) merge(a[u], q[u], k[u]);
  }
  for (; j < total; ++j) merge(base[((size_t)j * 3 + 0) * C], base[((size_t)j * 3 + 1) * C], base[((size_t)j * 3 + 2) * C]);
  const float n = nper * (float)total;
  const float var = fmaxf(m2 / n, 0.f), rstd = 1.f / sqrtf(var + eps);
  const float sc = gamma[c] * rstd, sh = beta[c];
  const size_t plane = (size_t)B * C;
  if (!batch_mode) {
    state[idx] = mean; state[plane + idx] = rstd; state[2 * plane + idx] = sc; state[3 * plane + idx] = sh;
  } else {
    for (int bb = 0; bb < B; ++bb) {
      const size_t o = (size_t)bb * C + c;
      state[o] = mean; state[plane + o] = rstd; state[2 * plane + o] = sc; state[3 * plane + o] = sh;
    }
    if (running_mean) {  // torch.nn.BatchNorm2d: unbiased variance in the running estimate
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * (n / fmaxf(n - 1.f, 1.f));
    }
  }
}


// InstanceNorm form of the above with the chunks of one (sample, channel) spread over 8 threads: block = 32 channels x 8 chunk groups,
// grid = (C/32, B).  Each thread merges its contiguous share of the chunks (equal sizes), the 8 group results are combined through LDS
// in group order with the general pairwise formula.  One thread per (b, c) walking 128 chunks through two dependent divisions each
// took 6.2 us on the critical chain 52 times per step.
__device__ __forceinline__ void norm_finalize_chunks_par_body(const float* __restrict__ partial, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float* __restrict__ state, int B, int HW,
                                                              int C, int chunks, float eps, int bx, int b) {
  __shared__ float smean[8][33], sm2[8][33];
  const int cl = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int c = bx * 32 + cl;
  const int per = chunks / 8;
  const float nper = (float)(HW / chunks), inv = 1.f / nper;
  float mean = 0.f, m2 = 0.f, cnt = 0.f;
  if (c < C) {
    const float* base = partial + (((size_t)b * chunks + (size_t)grp * per) * 3) * C + c;
    for (int j0 = 0; j0 < per; j0 += 8) {
      float a[8], q[8], k[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int j = min(j0 + u, per - 1);
        a[u] = base[((size_t)j * 3 + 0) * C]; q[u] = base[((size_t)j * 3 + 1) * C]; k[u] = base[((size_t)j * 3 + 2) * C];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (j0 + u < per) {
          const float mj = k[u] + a[u] * inv, d = mj - mean, c1 = cnt + 1.f;
          mean += d / c1;
          m2 += fmaxf(q[u] - a[u] * a[u] * inv, 0.f) + nper * d * d * (cnt / c1);
          cnt = c1;
        }
      }
    }
  }
  smean[grp][cl] = mean; sm2[grp][cl] = m2;
  __syncthreads();
  if (grp == 0 && c < C) {
    // groups hold equal counts n_g = per * nper: after merging t groups the running count is t * n_g
    float M = smean[0][cl], Q = sm2[0][cl];
    const float ng = (float)per * nper;
#pragma unroll
    for (int t = 1; t < 8; ++t) {
      const float d = smean[t][cl] - M, tf = (float)t;
      M += d / (tf + 1.f);
      Q += sm2[t][cl] + ng * d * d * (tf / (tf + 1.f));
    }
    const float n = (float)HW;
    const float var = fmaxf(Q / n, 0.f), rstd = 1.f / sqrtf(var + eps);
    const size_t plane = (size_t)B * C, idx = (size_t)b * C + c;
    state[idx] = M; state[plane + idx] = rstd; state[2 * plane + idx] = gamma[c] * rstd; state[3 * plane + idx] = beta[c];
  }
}
__global__ __launch_bounds__(256) void norm_finalize_chunks_par_kernel(const float* __restrict__ partial, const float* __restrict__ gamma,
                                                                       const float* __restrict__ beta, float* __restrict__ state, int B, int HW,
                                                                       int C, int chunks, float eps) {
  norm_finalize_chunks_par_body(partial, gamma, beta, state, B, HW, C, chunks, eps, blockIdx.x, blockIdx.y);
}
// two such finalisations of one shape in one launch (blockIdx.z picks the job): the two regression heads' norms of one depth (round 6: the
// heads' stretch of the forward runs alone on the chain, every launch boundary there is exposed)
struct FinalizePair { const float* partial[2]; const float* gamma[2]; const float* beta[2]; float* state[2]; };
__global__ __launch_bounds__(256) void norm_finalize_chunks_par_pair_kernel(FinalizePair q, int B, int HW, int C, int chunks, float eps) {
  const int z = blockIdx.z;
  norm_finalize_chunks_par_body(q.partial[z], q.gamma[z], q.beta[z], q.state[z], B, HW, C, chunks, eps, blockIdx.x, blockIdx.y);
}

}  // namespace pwr

extern "C" int pwr_norm_finalize_partial(const float* partial, int chunks, const float* gamma, const float* beta,
                                         float* running_mean, float* running_var, float* state, int B, int HW, int C, int mode,
                                         float eps, float momentum, void* stream) {
  if ((mode != 0 && mode != 1) || chunks < 1 || HW % chunks) return PWR_EINVAL;
  const int n = mode == 1 ? C : B * C;
  static const bool par = (PWR_DBG_ENV("PWR_NORM_PAR", 1) != 0);
  if (mode == 0 && par && chunks % 8 == 0 && chunks >= 16)
    hipLaunchKernelGGL(norm_finalize_chunks_par_kernel, dim3((C + 31) / 32, B), dim3(256), 0, (hipStream_t)stream, partial, gamma, beta, state, B,
                       HW, C, chunks, eps);
  else
    hipLaunchKernelGGL(norm_finalize_chunks_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, partial, gamma, beta, state,
                       running_mean, running_var, B, HW, C, chunks, mode, eps, momentum);
  return (int)hipGetLastError();
}

extern "C" int pwr_norm_finalize_partial_pair(const float* partial_a, const float* gamma_a, const float* beta_a, float* state_a,
                                              const float* partial_b, const float* gamma_b, const float* beta_b, float* state_b, int chunks,
                                              int B, int HW, int C, float eps, void* stream) {
  if (chunks < 1 || HW % chunks) return PWR_EINVAL;
  static const bool par = (PWR_DBG_ENV("PWR_NORM_PAR", 1) != 0);
  if (!(par && chunks % 8 == 0 && chunks >= 16)) {       // (the shapes pwr_norm_finalize_partial runs on its one-thread-per-channel kernel)
    int rc = pwr_norm_finalize_partial(partial_a, chunks, gamma_a, beta_a, nullptr, nullptr, state_a, B, HW, C, 0, eps, 0.1f, stream);
    if (rc) return rc;
    return pwr_norm_finalize_partial(partial_b, chunks, gamma_b, beta_b, nullptr, nullptr, state_b, B, HW, C, 0, eps, 0.1f, stream);
  }
  pwr::FinalizePair q{{partial_a, partial_b}, {gamma_a, gamma_b}, {beta_a, beta_b}, {state_a, state_b}};
  hipLaunchKernelGGL(pwr::norm_finalize_chunks_par_pair_kernel, dim3((C + 31) / 32, B, 2), dim3(256), 0, (hipStream_t)stream, q, B, HW, C, chunks, eps);
  return (int)hipGetLastError();
}

// Backward of relu(norm(y)) (relu optional).  g: upstream gradient; dy out (may alias g); addend optional (same
// shape, added to the result: the skip branch of a ResBlock).  S1,S2: [B,C] scratch.  dgamma/dbeta: [C].
// mode 2 (eval-mode batch norm) treats the statistics as constants.
extern "C" int pwr_norm_bwd(const void* g, const void* y, const float* state, float* partial, float* S1, float* S2, const void* addend, void* dy,
                            float* dgamma, float* dbeta, int accumulate, int relu, int B, int HW, int C, int mode, int dtype,
                            void* stream) {
  const int EP = dtype == PWR_BF16 ? 8 : 4;
  if (C % EP || C / EP > 256) return PWR_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const int nch = norm_chunks(B, HW);
  const int pl = 256 / (C / EP);
  const size_t sh = (size_t)pl * 2 * C * 4;
  partial = reinterpret_cast<float*>(reinterpret_cast<char*>(partial) + norm_counter_bytes(B));
  if (dtype == PWR_BF16) {
    hipLaunchKernelGGL((norm_bwd_partial_kernel<bf16_t>), dim3(nch, B), dim3(256), sh, s, (const bf16_t*)g, (const bf16_t*)y,
                       state, B, partial, HW, C, nch, relu);
  } else {
    hipLaunchKernelGGL((norm_bwd_partial_kernel<float>), dim3(nch, B), dim3(256), sh, s, (const float*)g, (const float*)y, state,
                       B, partial, HW, C, nch, relu);
  }
  hipLaunchKernelGGL(norm_bwd_sum_kernel, dim3((C + 7) / 8), dim3(256), 0, s, partial, S1, S2, dgamma, dbeta, B, HW, C, nch,
                     mode == 1 ? 1 : 0, accumulate);
  if (mode == 2) {  // statistics are constants: dy = scale * gm
    hipMemsetAsync(S1, 0, (size_t)B * C * 4, s);
    hipMemsetAsync(S2, 0, (size_t)B * C * 4, s);
  }
  if (dtype == PWR_BF16) {
    hipLaunchKernelGGL((norm_bwd_apply_kernel<bf16_t, false>), dim3(nch, B), dim3(256), 0, s, (const bf16_t*)g, (const bf16_t*)y, state, B,
                       S1, S2, (const bf16_t*)addend, (bf16_t*)dy, HW, C, nch, relu);
  } else {
    hipLaunchKernelGGL((norm_bwd_apply_kernel<float, false>), dim3(nch, B), dim3(256), 0, s, (const float*)g, (const float*)y, state, B,
                       S1, S2, (const float*)addend, (float*)dy, HW, C, nch, relu);
  }
  return (int)hipGetLastError();
}

// pwr_norm_bwd with the two reductions already done by the epilogue of the data-gradient conv that produced g
// (pwr_conv_fwd_stats, nb_partial: `chunks` slab rows per sample): 2 launches instead of 3, (g, y) read once instead of twice.
// Two pwr_norm_bwd_from_partial calls of one shape (instance norm, no addend) as two launches instead of four.  S1 / S2: 2 x B x C floats.
extern "C" int pwr_norm_bwd_from_partial_pair(const void* ga, const void* ya, const float* state_a, const float* partial_a, void* dya,
                                              float* dgamma_a, float* dbeta_a, const void* gb, const void* yb, const float* state_b,
                                              const float* partial_b, void* dyb, float* dgamma_b, float* dbeta_b, int chunks, float* S1,
                                              float* S2, int accumulate, int relu, int B, int HW, int C, int dtype, void* stream) {
  const int EP = dtype == PWR_BF16 ? 8 : 4;
  if (C % EP || C / EP > 256) return PWR_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const int nch = norm_chunks(B, HW);
  NormBwdPair q;
  q.j[0] = NormBwdJob{ga, ya, state_a, partial_a, S1, S2, dya, dgamma_a, dbeta_a};
  q.j[1] = NormBwdJob{gb, yb, state_b, partial_b, S1 + (size_t)B * C, S2 + (size_t)B * C, dyb, dgamma_b, dbeta_b};
  hipLaunchKernelGGL(norm_bwd_sum_pair_kernel, dim3((C + 7) / 8, 2), dim3(256), 0, s, q, B, HW, C, chunks, accumulate);
  if (dtype == PWR_BF16) hipLaunchKernelGGL((norm_bwd_apply_pair_kernel<bf16_t>), dim3(nch, B, 2), dim3(256), 0, s, q, B, HW, C, nch, relu);
  else hipLaunchKernelGGL((norm_bwd_apply_pair_kernel<float>), dim3(nch, B, 2), dim3(256), 0, s, q, B, HW, C, nch, relu);
  return (int)hipGetLastError();
}

extern "C" int pwr_norm_bwd_from_partial(const void* g, const void* y, const float* state, const float* partial, int chunks, float* S1,
                                         float* S2, const void* addend, void* dy, float* dgamma, float* dbeta, int accumulate, int relu,
                                         int B, int HW, int C, int mode, int dtype, void* stream) {
  const int EP = dtype == PWR_BF16 ? 8 : 4;
  if (C % EP || C / EP > 256 || mode == 2) return PWR_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const int nch = norm_chunks(B, HW);
  hipLaunchKernelGGL(norm_bwd_sum_kernel, dim3((C + 7) / 8), dim3(256), 0, s, partial, S1, S2, dgamma, dbeta, B, HW, C, chunks,
                     mode == 1 ? 1 : 0, accumulate);
  if (dtype == PWR_BF16) {
    hipLaunchKernelGGL((norm_bwd_apply_kernel<bf16_t, false>), dim3(nch, B), dim3(256), 0, s, (const bf16_t*)g, (const bf16_t*)y, state, B,
                       S1, S2, (const bf16_t*)addend, (bf16_t*)dy, HW, C, nch, relu);
  } else {
    hipLaunchKernelGGL((norm_bwd_apply_kernel<float, false>), dim3(nch, B), dim3(256), 0, s, (const float*)g, (const float*)y, state, B,
                       S1, S2, (const float*)addend, (float*)dy, HW, C, nch, relu);
  }
  return (int)hipGetLastError();
}

// Round 6 (instance norm): pwr_norm_bwd_from_partial as ONE launch on the caller's stream -- the apply kernel sums the slab rows of its
// sample itself (norm_bwd_apply_fold_body) -- for one tensor, or for the two heads' tensors of one depth (gb != NULL: blockIdx.z picks the
// job).  dy is bit-identical to pwr_norm_bwd_from_partial's.  The parameter gradients come from pwr_norm_bwd_params_from_partial on the
// same slab (any stream ordered behind the launch that wrote the slab).
extern "C" int pwr_norm_bwd_apply_from_partial(const void* ga, const void* ya, const float* state_a, const float* partial_a, const void* addend_a,
                                               void* dya, const void* gb, const void* yb, const float* state_b, const float* partial_b, void* dyb,
                                               int chunks, int relu, int B, int HW, int C, int dtype, void* stream) {
  const int EP = dtype == PWR_BF16 ? 8 : 4;
  if (C % EP || C / EP > 256 || 2 * C > 1024 || chunks < 1) return PWR_EUNSUPPORTED;
  if (!ga || !ya || !state_a || !partial_a || !dya || (gb && (!yb || !state_b || !partial_b || !dyb))) return PWR_EINVAL;
  // (the apply step's pixel chunks are a matter of launch shape only -- every element is computed alike -- so the fold launch may take
  // FEWER, larger workgroups than the two-launch form: each one pays the slab sums once.  PWR_FOLD_DIV, debug build: 1, 2, 4)
  static const int fdiv = PWR_DBG_ENV("PWR_FOLD_DIV", 1);
  int nch = norm_chunks(B, HW) / (fdiv > 0 ? fdiv : 1);
  if (nch < 1) nch = 1;
  NormBwdFoldPair q;
  q.j[0] = NormBwdFoldJob{ga, ya, state_a, partial_a, addend_a, dya};
  q.j[1] = NormBwdFoldJob{gb, yb, state_b, partial_b, nullptr, dyb};
  const dim3 grid(nch, B, gb ? 2 : 1);
  static const int ntl = PWR_DBG_ENV("PWR_FOLD_NT", 0);
  if (dtype == PWR_BF16 && ntl == 1) hipLaunchKernelGGL((norm_bwd_apply_fold_kernel<bf16_t, 1>), grid, dim3(256), 0, (hipStream_t)stream, q, B, HW, C, nch, chunks, relu);
  else if (dtype == PWR_BF16 && ntl == 3) hipLaunchKernelGGL((norm_bwd_apply_fold_kernel<bf16_t, 3>), grid, dim3(256), 0, (hipStream_t)stream, q, B, HW, C, nch, chunks, relu);
  else if (dtype == PWR_BF16) hipLaunchKernelGGL((norm_bwd_apply_fold_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, q, B, HW, C, nch, chunks, relu);
  else hipLaunchKernelGGL((norm_bwd_apply_fold_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, q, B, HW, C, nch, chunks, relu);
  return (int)hipGetLastError();
}

// dgamma / dbeta (+)= from the slab of pwr_conv_fwd_stats (nb_partial): the reduction launch of pwr_norm_bwd_from_partial without its
// S1 / S2 outputs (same kernel, same order, same bits); partial_b != NULL: a second job of the same shape in the same launch.
extern "C" int pwr_norm_bwd_params_from_partial(const float* partial_a, float* dgamma_a, float* dbeta_a, const float* partial_b, float* dgamma_b,
                                                float* dbeta_b, int chunks, int accumulate, int B, int HW, int C, void* stream) {
  if (!partial_a || !dgamma_a || !dbeta_a || chunks < 1 || (partial_b && (!dgamma_b || !dbeta_b))) return PWR_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (!partial_b) {
    hipLaunchKernelGGL(norm_bwd_sum_kernel, dim3((C + 7) / 8), dim3(256), 0, s, partial_a, (float*)nullptr, (float*)nullptr, dgamma_a, dbeta_a, B, HW,
                       C, chunks, 0, accumulate);
  } else {
    NormBwdPair q;
    q.j[0] = NormBwdJob{nullptr, nullptr, nullptr, partial_a, nullptr, nullptr, nullptr, dgamma_a, dbeta_a};
    q.j[1] = NormBwdJob{nullptr, nullptr, nullptr, partial_b, nullptr, nullptr, nullptr, dgamma_b, dbeta_b};
    hipLaunchKernelGGL(norm_bwd_sum_pair_kernel, dim3((C + 7) / 8, 2), dim3(256), 0, s, q, B, HW, C, chunks, accumulate);
  }
  return (int)hipGetLastError();
}

// The parameter gradients of `njobs` norm backwards (instance norm; each job = the slab of its data-gradient conv, pwr_conv_fwd_stats'
// nb_partial) in as few launches as a kernel-argument table allows (40 jobs each): bit-identical to pwr_norm_bwd_params_from_partial per job.
extern "C" int pwr_norm_bwd_params_group(const pwr_norm_param_job* jobs, int njobs, int accumulate, int B, void* stream) {
  if (njobs < 0 || (njobs && !jobs)) return PWR_EINVAL;
  for (int j0 = 0; j0 < njobs; j0 += pwr::NormParamGroup::kMax) {
    pwr::NormParamGroup g;
    g.n = njobs - j0 < pwr::NormParamGroup::kMax ? njobs - j0 : pwr::NormParamGroup::kMax;
    g.B = B; g.accumulate = accumulate;
    int blocks = 0;
    for (int j = 0; j < g.n; ++j) {
      const pwr_norm_param_job& q = jobs[j0 + j];
      if (!q.partial || !q.dgamma || !q.dbeta || q.chunks < 1 || q.C < 1) return PWR_EINVAL;
      g.partial[j] = q.partial; g.dgamma[j] = q.dgamma; g.dbeta[j] = q.dbeta; g.HW[j] = q.HW; g.C[j] = q.C; g.chunks[j] = q.chunks;
      g.first[j] = blocks;
      blocks += (q.C + 7) / 8;
    }
    g.first[g.n] = blocks;
    hipLaunchKernelGGL(pwr::norm_bwd_params_group_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g);
  }
  return (int)hipGetLastError();
}

// out [B,HW,C] = relu(norm(y)) (relu optional) in the activation dtype, from the [4][B][C] state: the operand the convs / weight gradients
// build on load, as a tensor (see norm_apply_kernel).
extern "C" int pwr_norm_apply(const void* y, const float* state, void* out, int relu, int B, int HW, int C, int dtype, void* stream) {
  const int EP = dtype == PWR_BF16 ? 8 : 4;
  if (C % EP || C / EP > 256 || !y || !state || !out) return PWR_EUNSUPPORTED;
  const int nch = norm_chunks(B, HW);
  if (dtype == PWR_BF16) hipLaunchKernelGGL((norm_apply_kernel<bf16_t>), dim3(nch, B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)y, state, (bf16_t*)out, B, HW, C, nch, relu);
  else hipLaunchKernelGGL((norm_apply_kernel<float>), dim3(nch, B), dim3(256), 0, (hipStream_t)stream, (const float*)y, state, (float*)out, B, HW, C, nch, relu);
  return (int)hipGetLastError();
}

// pwr_maxpool_fwd (src 1: xa [B,2H,2W,C]) or pwr_upsample_add_fwd (src 2: xa [B,H,W,C] skip, xh [B,H/2,W/2,C]) writing y [B,H,W,C], fused
// with pwr_norm_stats(y, ..., mode 0) of the InstanceNorm that follows: two launches (producer + partial sums, finalisation) instead of
// three, y and the [4][B][C] state bit-identical.  PWR_EUNSUPPORTED where pwr_norm_stats would take its one-block-per-sample form
// (H W <= 512) or the shape has no 16-byte channel slots: the caller then issues the separate launches.
extern "C" int pwr_norm_stats_fused_src(int src, const void* xa, const void* xh, void* y, const float* gamma, const float* beta, float* partial,
                                        float* state, int B, int H, int W, int C, float eps, int dtype, void* stream) {
  const int EP = dtype == PWR_BF16 ? 8 : 4, HW = H * W;
  static const int fwd_small = PWR_DBG_ENV("PWR_NORM_FWD_SMALL", 512);
  if (C % EP || C / EP > 256 || HW <= fwd_small || (src != 1 && src != 2) || (src == 2 && ((H | W) & 1))) return PWR_EUNSUPPORTED;
  if (!xa || (src == 2 && !xh) || !y || !partial || !state) return PWR_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int nch = norm_chunks(B, HW);
  const int pl = 256 / (C / EP);
  const size_t sh = (size_t)pl * 2 * C * 4;
  partial = reinterpret_cast<float*>(reinterpret_cast<char*>(partial) + norm_counter_bytes(B));
  if (dtype == PWR_BF16) {
    if (src == 1) hipLaunchKernelGGL((norm_partial_src_kernel<bf16_t, 1>), dim3(nch, B), dim3(256), sh, s, (const bf16_t*)xa, (const bf16_t*)xh, (bf16_t*)y, partial, H, W, C, nch);
    else hipLaunchKernelGGL((norm_partial_src_kernel<bf16_t, 2>), dim3(nch, B), dim3(256), sh, s, (const bf16_t*)xa, (const bf16_t*)xh, (bf16_t*)y, partial, H, W, C, nch);
    hipLaunchKernelGGL((norm_finalize_kernel<bf16_t>), dim3((B * C + 255) / 256), dim3(256), 0, s, (const bf16_t*)y, partial, gamma, beta, state, nullptr,
                       nullptr, B, HW, C, nch, 0, eps, 0.1f);
  } else {
    if (src == 1) hipLaunchKernelGGL((norm_partial_src_kernel<float, 1>), dim3(nch, B), dim3(256), sh, s, (const float*)xa, (const float*)xh, (float*)y, partial, H, W, C, nch);
    else hipLaunchKernelGGL((norm_partial_src_kernel<float, 2>), dim3(nch, B), dim3(256), sh, s, (const float*)xa, (const float*)xh, (float*)y, partial, H, W, C, nch);
    hipLaunchKernelGGL((norm_finalize_kernel<float>), dim3((B * C + 255) / 256), dim3(256), 0, s, (const float*)y, partial, gamma, beta, state, nullptr,
                       nullptr, B, HW, C, nch, 0, eps, 0.1f);
  }
  return (int)hipGetLastError();
}
