// Convolutions that are not GEMM-shaped enough for the matrix cores (HBM-bound, done on the VALU):
//   * stem conv Cin = 1 (model.py:165):            img [B,S,S] fp32 -> y [B,S,S,C0] T, k x k, stride 1
//   * stage-input 1x1 conv on the NCHW concat (model.py:137 applied to model.py:208):
//       cat[heatmaps(J), depthmaps(J), label_img(1)] is never materialised -- the kernel reads the three
//       fp32 NCHW sources directly and writes NHWC T.
// plus their gradients.  Partial sums go to slabs reduced in a fixed order (deterministic).
#include "pwr_common.h"
#include "pwr.h"

namespace pwr {

// ------------------------------------------------------------------ stem conv, Cin = 1
template <typename T, int KS>
__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                       const float* __restrict__ bias, T* __restrict__ y, int B, int S,
                                                       int C0) {
  constexpr int ks = KS;
  constexpr int EP = Elem<T>::kPer16B;
  typedef typename Vec16<T>::type V;
  extern __shared__ float sw[];  // [C0][ks*ks] + [C0]
  const int taps = ks * ks, pad = ks / 2;
  for (int i = threadIdx.x; i < C0 * taps; i += 256) sw[i] = w[i];
  for (int i = threadIdx.x; i < C0; i += 256) sw[C0 * taps + i] = bias[i];
  __syncthreads();
  const int cpp = C0 / EP;  // chunks per pixel
  const long long total = (long long)B * S * S * cpp;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int cq = (int)(idx % cpp);
    const long long pix = idx / cpp;
    const int xx = (int)(pix % S), yy = (int)((pix / S) % S), b = (int)(pix / ((long long)S * S));
    float in[KS * KS];
#pragma unroll
    for (int ky = 0; ky < ks; ++ky)
#pragma unroll
      for (int kx = 0; kx < ks; ++kx) {
        const int iy = yy + ky - pad, ix = xx + kx - pad;
        in[ky * ks + kx] = (iy >= 0 && iy < S && ix >= 0 && ix < S) ? img[((size_t)b * S + iy) * S + ix] : 0.f;
      }
    V o;
#pragma unroll
    for (int e = 0; e < EP; ++e) {
      const int c = cq * EP + e;
      float a = 0.f;
#pragma unroll
      for (int t = 0; t < KS * KS; ++t) a = fmaf(in[t], sw[c * taps + t], a);
      o[e] = Elem<T>::from_f(a + sw[C0 * taps + c]);
    }
    *reinterpret_cast<V*>(y + (size_t)pix * C0 + cq * EP) = o;
  }
}

// The same conv for 3x3, C0 = 32 (4 chunks of 8 bf16 channels), S % 128 == 0 (round 6): a workgroup owns 4 rows x 128 columns.  The 6 x 130
// image window goes to LDS once (zero padding included), a thread keeps the 72 weights + 8 biases of ITS channel chunk in registers and
// walks 8 pixels (pass i: pixel 64 i + tid / 4 of the tile), so that a store instruction of a wave writes 16 consecutive pixels = 1 KiB.
// Per output vector: 9 LDS reads + 72 FMAs against 72 LDS reads + 72 FMAs + 9 bounds-checked global loads + 64-bit index arithmetic
// above.  Same FMA order per channel (taps ascending, then the bias): same bits.
template <typename T>
__global__ __launch_bounds__(256) void stem_fwd_tile_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                            const float* __restrict__ bias, T* __restrict__ y, int B, int S) {
  constexpr int EP = Elem<T>::kPer16B, C0 = 32, CPP = C0 / EP, PPW = 256 / CPP, TW = 128, TH = 4, LW = TW + 2;
  typedef typename Vec16<T>::type V;
  __shared__ float simg[(TH + 2) * LW];
  __shared__ float sw[C0 * 9 + C0];
  const int tid = threadIdx.x;
  const int tiles_x = S / TW, tiles_y = S / TH;
  const int t = blockIdx.x, b = t / (tiles_x * tiles_y), tr = t - b * tiles_x * tiles_y;
  const int y0 = (tr / tiles_x) * TH, x0 = (tr % tiles_x) * TW;
  for (int i = tid; i < C0 * 9; i += 256) sw[i] = w[i];
  if (tid < C0) sw[C0 * 9 + tid] = bias[tid];
  for (int i = tid; i < (TH + 2) * LW; i += 256) {
    const int r = i / LW, c = i - r * LW;
    const int iy = y0 + r - 1, ix = x0 + c - 1;
    simg[i] = (iy >= 0 && iy < S && ix >= 0 && ix < S) ? img[((size_t)b * S + iy) * S + ix] : 0.f;
  }
  __syncthreads();
  const int cq = tid % CPP, pl = tid / CPP;
  float wr[EP][9], br[EP];
#pragma unroll
  for (int e = 0; e < EP; ++e) {
#pragma unroll
    for (int k = 0; k < 9; ++k) wr[e][k] = sw[(cq * EP + e) * 9 + k];
    br[e] = sw[C0 * 9 + cq * EP + e];
  }
#pragma unroll
  for (int i = 0; i < TW * TH / PPW; ++i) {
    const int p = i * PPW + pl, py = p / TW, px = p - py * TW;
    float in[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) in[ky * 3 + kx] = simg[(py + ky) * LW + px + kx];
    V o;
#pragma unroll
    for (int e = 0; e < EP; ++e) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < 9; ++k) a = fmaf(in[k], wr[e][k], a);
      o[e] = Elem<T>::from_f(a + br[e]);
    }
    *reinterpret_cast<V*>(y + (((size_t)b * S + y0 + py) * S + x0 + px) * C0 + cq * EP) = o;
  }
}

// dW[c][tap] partials: slab[block][c*taps + tap].  A thread owns one 16-byte channel slot of dy for every PLN-th pixel of the
// block's pixel range (coalesced 16-byte loads, coordinates stepped incrementally: no divisions in the loop); the 3x3 (k x k)
// image neighbourhood is loaded once per pixel and shared by the slot's channels.  Pixel lanes are combined with wave shuffles,
// the four waves through LDS, in a fixed order.
// KR kernel rows per pass (blockIdx.y = pass): 7x7 keeps 3 rows = 21 taps x 8 channels of accumulators in registers, not 49 x 8.
template <typename T, int KS, int KR>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ img, const T* __restrict__ dy,
                                                         float* __restrict__ slab, int B, int S, int C0,
                                                         int pix_per_block) {
  constexpr int TT = KS * KS, TP = KR * KS, pad = KS / 2, EP = Elem<T>::kPer16B;
  typedef typename Vec16<T>::type V;
  extern __shared__ float red[];           // [4 waves][C0][TP]
  const int ky0 = blockIdx.y * KR;
  const int NSL = C0 / EP, PLN = 256 / NSL;          // channel slots, pixel lanes (C0 / EP divides 64)
  const int slot = threadIdx.x % NSL, pl = threadIdx.x / NSL;
  const long long npix = (long long)B * S * S;
  const long long p0 = (long long)blockIdx.x * pix_per_block;
  long long pend = p0 + pix_per_block;
  if (pend > npix) pend = npix;
  float acc[EP][TP];
#pragma unroll
  for (int e = 0; e < EP; ++e)
#pragma unroll
    for (int t = 0; t < TP; ++t) acc[e][t] = 0.f;
  long long pp = p0 + pl;
  int x = (int)(pp % S), y = (int)((pp / S) % S), b = (int)(pp / ((long long)S * S));
#pragma unroll 2
  for (; pp < pend; pp += PLN) {
    const V g = *reinterpret_cast<const V*>(dy + (size_t)pp * C0 + slot * EP);
    float v[TP];
    const float* ib = img + (size_t)b * S * S;
#pragma unroll
    for (int ky = 0; ky < KR; ++ky)
#pragma unroll
      for (int kx = 0; kx < KS; ++kx) {
        const int iy = y + ky0 + ky - pad, ix = x + kx - pad;
        const bool ok = ky0 + ky < KS && iy >= 0 && iy < S && ix >= 0 && ix < S;
        const float t = ib[ok ? iy * S + ix : 0];
        v[ky * KS + kx] = ok ? t : 0.f;
      }
#pragma unroll
    for (int e = 0; e < EP; ++e) {
      const float ge = Elem<T>::to_f(g[e]);
#pragma unroll
      for (int t = 0; t < TP; ++t) acc[e][t] = fmaf(ge, v[t], acc[e][t]);
    }
    x += PLN;
    while (x >= S) { x -= S; if (++y == S) { y = 0; ++b; } }
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int e = 0; e < EP; ++e)
#pragma unroll
    for (int t = 0; t < TP; ++t) {
      float a = acc[e][t];
      // (lane ^ o for o = NSL ... 32 -- NSL, the channel slots, is a run-time power of two -- by DPP / v_permlane swaps, pwr_common.h: the sums of
      // the __shfl_xor loop; the conditions are wave-uniform)
      if (NSL <= 1) a = lane_xor_add<1>(a);
      if (NSL <= 2) a = lane_xor_add<2>(a);
      if (NSL <= 4) a = lane_xor_add<4>(a);
      if (NSL <= 8) a = lane_xor_add<8>(a);
      if (NSL <= 16) a = lane_xor_add<16>(a);
      if (NSL <= 32) a = lane_xor_add<32>(a);
      if (lane < NSL) red[(wid * C0 + slot * EP + e) * TP + t] = a;
    }
  __syncthreads();
  const int ntp = (KS - ky0 < KR ? KS - ky0 : KR) * KS;      // taps of this pass
  for (int i = threadIdx.x; i < C0 * TP; i += 256) {
    const int c = i / TP, tp = i - c * TP;
    if (tp < ntp)
      slab[(size_t)blockIdx.x * C0 * TT + c * TT + ky0 * KS + tp] = (red[i] + red[C0 * TP + i]) + (red[2 * C0 * TP + i] + red[3 * C0 * TP + i]);
  }
}

// out[i] (+)= sum_s slab[s*n + i]
__global__ void slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out, int S, int n, int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  int k = 0;
  // (32 loads in flight first: the stem's weight gradient sums 512 slabs on 2 blocks at the very end of the train step -- 64 dependent
  // L2 round trips = 22 us with 8 in flight; same ascending order of additions, so the same bits)
  for (; k + 32 <= S; k += 32) {
    float a[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) a[u] = slab[(size_t)(k + u) * n + i];
#pragma unroll
    for (int u = 0; u < 32; ++u) s += a[u];
  }
  for (; k + 8 <= S; k += 8) {      // 8 loads in flight, fixed summation order
    float a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = slab[(size_t)(k + u) * n + i];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += a[u];
  }
  for (; k < S; ++k) s += slab[(size_t)k * n + i];
  out[i] = accumulate ? out[i] + s : s;
}

// ------------------------------------------------------------------ stage-input 1x1 on NCHW sources
// channel c: c < J -> pmap[b,c], c < 2J -> dmap[b,c-J], c == 2J -> label[b]
__device__ __forceinline__ float cat_load(const float* pmap, const float* dmap, const float* label, int b, int c,
                                          int J, int N, int pix) {
  if (c < J) return pmap[((size_t)b * J + c) * N + pix];
  if (c < 2 * J) return dmap[((size_t)b * J + (c - J)) * N + pix];
  return label[(size_t)b * N + pix];
}

template <typename T>
__global__ __launch_bounds__(256) void catconv_fwd_kernel(const float* __restrict__ pmap, const float* __restrict__ dmap,
                                                          const float* __restrict__ label, const float* __restrict__ w,
                                                          const float* __restrict__ bias, T* __restrict__ y, int B, int N,
                                                          int J, int F) {
  constexpr int EP = Elem<T>::kPer16B;
  typedef typename Vec16<T>::type V;
  constexpr int TP = 64;  // pixels per block
  extern __shared__ float sm[];
  const int K = 2 * J + 1;
  float* sw = sm;                 // [K][F]  (transposed from [F][K])
  float* sb = sw + K * F;         // [F]
  float* sx = sb + F;             // [K][TP]
  for (int i = threadIdx.x; i < K * F; i += 256) { const int f = i / K, k = i - f * K; sw[k * F + f] = w[i]; }
  for (int i = threadIdx.x; i < F; i += 256) sb[i] = bias[i];
  const long long m0 = (long long)blockIdx.x * TP, M = (long long)B * N;
  for (int i = threadIdx.x; i < K * TP; i += 256) {
    const int k = i / TP, pp = i - k * TP;
    const long long m = m0 + pp;
    float v = 0.f;
    if (m < M) { const int b = (int)(m / N), pix = (int)(m - (long long)b * N); v = cat_load(pmap, dmap, label, b, k, J, N, pix); }
    sx[i] = v;
  }
  __syncthreads();
  const int cpp = F / EP;
  for (int c = threadIdx.x; c < TP * cpp; c += 256) {
    const int pp = c / cpp, cq = c - pp * cpp;
    const long long m = m0 + pp;
    if (m >= M) continue;
    float a[EP];
#pragma unroll
    for (int e = 0; e < EP; ++e) a[e] = sb[cq * EP + e];
    for (int k = 0; k < K; ++k) {
      const float xv = sx[k * TP + pp];
#pragma unroll
      for (int e = 0; e < EP; ++e) a[e] = fmaf(xv, sw[k * F + cq * EP + e], a[e]);
    }
    V o;
#pragma unroll
    for (int e = 0; e < EP; ++e) o[e] = Elem<T>::from_f(a[e]);
    *reinterpret_cast<V*>(y + (size_t)m * F + cq * EP) = o;
  }
}

// dgrad: g[b,c,pix] = sum_f W[f][c] * dy[b,pix,f]  for c < 2J  -> gp [B,J,N], gd [B,J,N] fp32 (overwrite)
template <typename T>
__global__ __launch_bounds__(256) void catconv_dgrad_kernel(const T* __restrict__ dy, const float* __restrict__ w,
                                                            float* __restrict__ gp, float* __restrict__ gd, int B, int N,
                                                            int J, int F) {
  constexpr int TP = 64;
  extern __shared__ float sm[];
  const int K = 2 * J + 1, FP = F + 1;
  float* sw = sm;              // [K][F]
  float* sd = sw + K * F;      // [TP][F+1]
  for (int i = threadIdx.x; i < K * F; i += 256) { const int f = i / K, k = i - f * K; sw[k * F + f] = w[i]; }
  const long long m0 = (long long)blockIdx.x * TP, M = (long long)B * N;
  for (int i = threadIdx.x; i < TP * F; i += 256) {
    const int pp = i / F, f = i - pp * F;
    const long long m = m0 + pp;
    sd[pp * FP + f] = (m < M) ? Elem<T>::to_f(dy[(size_t)m * F + f]) : 0.f;
  }
  __syncthreads();
  const int pp = threadIdx.x % TP, cg = threadIdx.x / TP;   // 4 channel groups
  const long long m = m0 + pp;
  if (m >= M) return;
  const int b = (int)(m / N), pix = (int)(m - (long long)b * N);
  for (int c = cg; c < 2 * J; c += 256 / TP) {
    float a = 0.f;
    for (int f = 0; f < F; ++f) a = fmaf(sw[c * F + f], sd[pp * FP + f], a);
    if (c < J) gp[((size_t)b * J + c) * N + pix] = a;
    else gd[((size_t)b * J + (c - J)) * N + pix] = a;
  }
}

// wgrad partials: slab[block][(K+1)][F]: rows k<K: sum dy*x_k ; row K: sum dy (bias grad)
template <typename T>
__global__ __launch_bounds__(256) void catconv_wgrad_kernel(const float* __restrict__ pmap, const float* __restrict__ dmap,
                                                            const float* __restrict__ label, const T* __restrict__ dy,
                                                            float* __restrict__ slab, int B, int N, int J, int F,
                                                            int pix_per_block) {
  constexpr int TP = 64;
  extern __shared__ float sx[];  // [K+1][TP]
  const int K = 2 * J + 1, K1 = K + 1;
  const long long M = (long long)B * N;
  const long long p0 = (long long)blockIdx.x * pix_per_block;
  // thread -> output channel f = tid (F <= 256); the K+1 rows are accumulated in registers
  constexpr int KMAX = 96;   // 2J+2 <= 96  (J <= 47)
  const int f = threadIdx.x;
  const bool act = f < F;
  float acc[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) acc[k] = 0.f;
  for (long long t0 = p0; t0 < p0 + pix_per_block && t0 < M; t0 += TP) {
    __syncthreads();
    for (int i = threadIdx.x; i < K1 * TP; i += 256) {
      const int k = i / TP, pp = i - k * TP;
      const long long m = t0 + pp;
      float v = 0.f;
      if (m < M && m < p0 + pix_per_block) {
        const int b = (int)(m / N), pix = (int)(m - (long long)b * N);
        v = (k < K) ? cat_load(pmap, dmap, label, b, k, J, N, pix) : 1.f;
      }
      sx[i] = v;
    }
    __syncthreads();
    if (act) {
      for (int pp = 0; pp < TP; ++pp) {
        const long long m = t0 + pp;
        if (m >= M || m >= p0 + pix_per_block) break;
        const float g = Elem<T>::to_f(dy[(size_t)m * F + f]);
#pragma unroll
        for (int kb = 0; kb < KMAX; kb += 8) {
          if (kb < K1) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
              if (kb + k < K1) acc[kb + k] = fmaf(g, sx[(kb + k) * TP + pp], acc[kb + k]);
          }
        }
      }
    }
  }
  if (act) {
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < K1) slab[((size_t)blockIdx.x * K1 + k) * F + f] = acc[k];
  }
}

// scatter reduced [K+1][F] -> dW [F][K] (OIHW with 1x1) and db [F]
__global__ void catconv_wgrad_finish(const float* __restrict__ slab, float* __restrict__ dw, float* __restrict__ db, int S,
                                     int K, int F, int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (K + 1) * F) return;
  const int k = i / F, f = i - k * F;
  float s = 0.f;
  for (int j = 0; j < S; ++j) s += slab[(size_t)j * (K + 1) * F + i];
  float* dst = (k < K) ? &dw[(size_t)f * K + k] : &db[f];
  *dst = accumulate ? *dst + s : s;
}

}  // namespace pwr

using namespace pwr;

extern "C" int pwr_stem_conv_fwd(const float* img, const float* w, const float* bias, void* y, int B, int S, int C0,
                                 int ksize, int dtype, void* stream) {
  if (C0 % 8 || C0 > 256) return PWR_EUNSUPPORTED;
  const int EP = dtype == PWR_BF16 ? 8 : 4;
  const long long total = (long long)B * S * S * (C0 / EP);
  const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  const size_t sh = (size_t)(C0 * ksize * ksize + C0) * 4;
  static const bool tile_on = PWR_DBG_ENV("PWR_STEM_TILE", 1) != 0;
  if (tile_on && dtype == PWR_BF16 && ksize == 3 && C0 == 32 && S % 128 == 0) {
    hipLaunchKernelGGL((stem_fwd_tile_kernel<bf16_t>), dim3(B * (S / 128) * (S / 4)), dim3(256), 0, (hipStream_t)stream, img, w, bias, (bf16_t*)y, B, S);
    return (int)hipGetLastError();
  }
#define PWR_STEM_F(KS_) \
  if (dtype == PWR_BF16) hipLaunchKernelGGL((stem_fwd_kernel<bf16_t, KS_>), dim3(grid), dim3(256), sh, (hipStream_t)stream, img, w, bias, (bf16_t*)y, B, S, C0); \
  else hipLaunchKernelGGL((stem_fwd_kernel<float, KS_>), dim3(grid), dim3(256), sh, (hipStream_t)stream, img, w, bias, (float*)y, B, S, C0)
  if (ksize == 1) { PWR_STEM_F(1); } else if (ksize == 3) { PWR_STEM_F(3); } else if (ksize == 5) { PWR_STEM_F(5); } else if (ksize == 7) { PWR_STEM_F(7); } else return PWR_EUNSUPPORTED;
  return (int)hipGetLastError();
}

extern "C" int pwr_stem_conv_wgrad_blocks(int B, int S) {
  long long npix = (long long)B * S * S;
  long long nb = (npix + 1023) / 1024;
  return (int)(nb > 1024 ? 1024 : nb);
}

extern "C" int pwr_stem_conv_wgrad(const float* img, const void* dy, float* slab, float* dw, int accumulate, int B, int S,
                                   int C0, int ksize, int dtype, void* stream) {
  const int EPh = dtype == PWR_BF16 ? 8 : 4;
  if (C0 % EPh || 64 % (C0 / EPh)) return PWR_EUNSUPPORTED;
  const int kr = ksize == 7 ? 3 : ksize, passes = (ksize + kr - 1) / kr;
  const size_t shw = (size_t)4 * C0 * kr * ksize * sizeof(float);
  const int nb = pwr_stem_conv_wgrad_blocks(B, S);
  const long long npix = (long long)B * S * S;
  const int ppb = (int)((npix + nb - 1) / nb);
  hipStream_t s = (hipStream_t)stream;
#define PWR_STEM_W(KS_, KR_) \
  if (dtype == PWR_BF16) hipLaunchKernelGGL((stem_wgrad_kernel<bf16_t, KS_, KR_>), dim3(nb, passes), dim3(256), shw, s, img, (const bf16_t*)dy, slab, B, S, C0, ppb); \
  else hipLaunchKernelGGL((stem_wgrad_kernel<float, KS_, KR_>), dim3(nb, passes), dim3(256), shw, s, img, (const float*)dy, slab, B, S, C0, ppb)
  if (ksize == 1) { PWR_STEM_W(1, 1); } else if (ksize == 3) { PWR_STEM_W(3, 3); } else if (ksize == 5) { PWR_STEM_W(5, 5); }
  else if (ksize == 7) { PWR_STEM_W(7, 3); } else return PWR_EUNSUPPORTED;
  const int n = C0 * ksize * ksize;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, s, slab, dw, nb, n, accumulate);
  return (int)hipGetLastError();
}

extern "C" int pwr_slab_reduce(const float* slab, float* out, int S, int n, int accumulate, void* stream) {
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, slab, out, S, n, accumulate);
  return (int)hipGetLastError();
}

extern "C" int pwr_catconv_fwd(const float* pmap, const float* dmap, const float* label, const float* w, const float* bias,
                               void* y, int B, int N, int J, int F, int dtype, void* stream) {
  const int K = 2 * J + 1;
  if (F % 8) return PWR_EUNSUPPORTED;
  const size_t sh = (size_t)(K * F + F + K * 64) * 4;
  if (sh > 160 * 1024) return PWR_EUNSUPPORTED;
  const long long M = (long long)B * N;
  const int grid = (int)((M + 63) / 64);
  if (dtype == PWR_BF16) hipLaunchKernelGGL((catconv_fwd_kernel<bf16_t>), dim3(grid), dim3(256), sh, (hipStream_t)stream, pmap, dmap, label, w, bias, (bf16_t*)y, B, N, J, F);
  else hipLaunchKernelGGL((catconv_fwd_kernel<float>), dim3(grid), dim3(256), sh, (hipStream_t)stream, pmap, dmap, label, w, bias, (float*)y, B, N, J, F);
  return (int)hipGetLastError();
}

extern "C" int pwr_catconv_dgrad(const void* dy, const float* w, float* gp, float* gd, int B, int N, int J, int F,
                                 int dtype, void* stream) {
  const int K = 2 * J + 1;
  const size_t sh = (size_t)(K * F + 64 * (F + 1)) * 4;
  if (sh > 160 * 1024) return PWR_EUNSUPPORTED;
  const long long M = (long long)B * N;
  const int grid = (int)((M + 63) / 64);
  if (dtype == PWR_BF16) hipLaunchKernelGGL((catconv_dgrad_kernel<bf16_t>), dim3(grid), dim3(256), sh, (hipStream_t)stream, (const bf16_t*)dy, w, gp, gd, B, N, J, F);
  else hipLaunchKernelGGL((catconv_dgrad_kernel<float>), dim3(grid), dim3(256), sh, (hipStream_t)stream, (const float*)dy, w, gp, gd, B, N, J, F);
  return (int)hipGetLastError();
}

extern "C" int pwr_catconv_wgrad_blocks(int B, int N) {
  long long M = (long long)B * N;
  long long nb = (M + 511) / 512;
  return (int)(nb > 512 ? 512 : nb);
}

extern "C" int pwr_catconv_wgrad(const float* pmap, const float* dmap, const float* label, const void* dy, float* slab,
                                 float* dw, float* db, int accumulate, int B, int N, int J, int F, int dtype, void* stream) {
  const int K = 2 * J + 1;
  if (K + 1 > 96 || F > 256) return PWR_EUNSUPPORTED;
  const int nb = pwr_catconv_wgrad_blocks(B, N);
  const long long M = (long long)B * N;
  int ppb = (int)((M + nb - 1) / nb);
  ppb = (ppb + 63) / 64 * 64;
  const size_t sh = (size_t)(K + 1) * 64 * 4;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == PWR_BF16) hipLaunchKernelGGL((catconv_wgrad_kernel<bf16_t>), dim3(nb), dim3(256), sh, s, pmap, dmap, label, (const bf16_t*)dy, slab, B, N, J, F, ppb);
  else hipLaunchKernelGGL((catconv_wgrad_kernel<float>), dim3(nb), dim3(256), sh, s, pmap, dmap, label, (const float*)dy, slab, B, N, J, F, ppb);
  const int n = (K + 1) * F;
  hipLaunchKernelGGL(catconv_wgrad_finish, dim3((n + 255) / 256), dim3(256), 0, s, slab, dw, db, nb, K, F, accumulate);
  return (int)hipGetLastError();
}
