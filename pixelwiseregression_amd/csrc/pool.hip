// Hourglass plumbing on NHWC activations (all HBM-bound, 16-byte vectors per lane):
//   MaxPool2d(2, stride=2)                                   model.py:29,40
//   F.interpolate(h, size=x.shape[2:]) (nearest) + skip add   model.py:45-47
// and their gradients; plus the NCHW fp32 -> NHWC T (zero-padded channels) transpose that feeds the
// decoder's logits/depth gradients into the head convolutions' backward.
#include "pwr_common.h"
#include "pwr.h"

namespace pwr {

__device__ __forceinline__ int nearest_src(int dst, int in, int out) {
  // ATen nearest_neighbor_compute_source_index with scale = (float)in/out
  const float scale = (float)in / (float)out;
  const int s = (int)floorf((float)dst * scale);
  return s < in - 1 ? s : in - 1;
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C) {
  constexpr int EP = Elem<T>::kPer16B;
  typedef typename Vec16<T>::type V;
  const int Ho = H / 2, Wo = W / 2, cpp = C / EP;
  const long long total = (long long)B * Ho * Wo * cpp;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int cq = (int)(i % cpp);
    long long r = i / cpp;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho);
    const int b = (int)(r / Ho);
    const T* p = x + (((size_t)b * H + 2 * oy) * W + 2 * ox) * C + cq * EP;
    V v00 = *reinterpret_cast<const V*>(p), v01 = *reinterpret_cast<const V*>(p + C);
    V v10 = *reinterpret_cast<const V*>(p + (size_t)W * C), v11 = *reinterpret_cast<const V*>(p + (size_t)W * C + C);
    V o;
#pragma unroll
    for (int e = 0; e < EP; ++e) {
      const float m = fmaxf(fmaxf(Elem<T>::to_f(v00[e]), Elem<T>::to_f(v01[e])), fmaxf(Elem<T>::to_f(v10[e]), Elem<T>::to_f(v11[e])));
      o[e] = Elem<T>::from_f(m);
    }
    *reinterpret_cast<V*>(y + (((size_t)b * Ho + oy) * Wo + ox) * C + cq * EP) = o;
  }
}

// dx = addend + route(dh): the gradient goes to the first maximum of each 2x2 window in scan order
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dh,
                                                          const T* __restrict__ addend, T* __restrict__ dx, int B, int H, int W,
                                                          int C) {
  constexpr int EP = Elem<T>::kPer16B;
  typedef typename Vec16<T>::type V;
  const int Ho = H / 2, Wo = W / 2, cpp = C / EP;
  const int Hc = (H + 1) / 2, Wc = (W + 1) / 2;   // windows incl. the ragged edge (which has no pool output)
  const long long total = (long long)B * Hc * Wc * cpp;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int cq = (int)(i % cpp);
    long long r = i / cpp;
    const int ox = (int)(r % Wc); r /= Wc;
    const int oy = (int)(r % Hc);
    const int b = (int)(r / Hc);
    const bool pooled = oy < Ho && ox < Wo;
    V xv[4], ov[4];
    bool ok[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int yy = 2 * oy + (t >> 1), xx = 2 * ox + (t & 1);
      ok[t] = yy < H && xx < W;
      const size_t off = (((size_t)b * H + yy) * W + xx) * C + cq * EP;
      if (ok[t]) {
        xv[t] = *reinterpret_cast<const V*>(x + off);
        if (addend) ov[t] = *reinterpret_cast<const V*>(addend + off);
        else ov[t] = V{};
      }
    }
    if (pooled) {
      V g = *reinterpret_cast<const V*>(dh + (((size_t)b * Ho + oy) * Wo + ox) * C + cq * EP);
#pragma unroll
      for (int e = 0; e < EP; ++e) {
        int best = 0;
        float bv = Elem<T>::to_f(xv[0][e]);
#pragma unroll
        for (int t = 1; t < 4; ++t) {
          const float v = Elem<T>::to_f(xv[t][e]);
          if (v > bv) { bv = v; best = t; }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
          if (t == best) ov[t][e] = Elem<T>::from_f(Elem<T>::to_f(ov[t][e]) + Elem<T>::to_f(g[e]));
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int yy = 2 * oy + (t >> 1), xx = 2 * ox + (t & 1);
      if (ok[t]) *reinterpret_cast<V*>(dx + (((size_t)b * H + yy) * W + xx) * C + cq * EP) = ov[t];
    }
  }
}

// out[b,y,x,:] = skip[b,y,x,:] + h[b,sy(y),sx(x),:]
template <typename T>
__global__ __launch_bounds__(256) void upsample_add_kernel(const T* __restrict__ h, const T* __restrict__ skip, T* __restrict__ out,
                                                           int B, int Hi, int Wi, int Ho, int Wo, int C) {
  constexpr int EP = Elem<T>::kPer16B;
  typedef typename Vec16<T>::type V;
  const int cpp = C / EP;
  const long long total = (long long)B * Ho * Wo * cpp;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int cq = (int)(i % cpp);
    long long r = i / cpp;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho);
    const int b = (int)(r / Ho);
    const int sy = nearest_src(oy, Hi, Ho), sx = nearest_src(ox, Wi, Wo);
    V hv = *reinterpret_cast<const V*>(h + (((size_t)b * Hi + sy) * Wi + sx) * C + cq * EP);
    const size_t off = (((size_t)b * Ho + oy) * Wo + ox) * C + cq * EP;
    V sv = *reinterpret_cast<const V*>(skip + off);
    V o;
#pragma unroll
    for (int e = 0; e < EP; ++e) o[e] = Elem<T>::from_f(Elem<T>::to_f(hv[e]) + Elem<T>::to_f(sv[e]));
    *reinterpret_cast<V*>(out + off) = o;
  }
}

// dh[b,hy,hx,:] = sum over the output pixels that read (hy,hx)
template <typename T>
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const T* __restrict__ dout, T* __restrict__ dh, int B, int Hi, int Wi,
                                                           int Ho, int Wo, int C) {
  constexpr int EP = Elem<T>::kPer16B;
  typedef typename Vec16<T>::type V;
  const int cpp = C / EP;
  const long long total = (long long)B * Hi * Wi * cpp;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int cq = (int)(i % cpp);
    long long r = i / cpp;
    const int hx = (int)(r % Wi); r /= Wi;
    const int hy = (int)(r % Hi);
    const int b = (int)(r / Hi);
    float a[EP];
#pragma unroll
    for (int e = 0; e < EP; ++e) a[e] = 0.f;
    const int y0 = max(0, (int)((long long)hy * Ho / Hi) - 1), y1 = min(Ho - 1, (int)((long long)(hy + 1) * Ho / Hi) + 1);
    const int x0 = max(0, (int)((long long)hx * Wo / Wi) - 1), x1 = min(Wo - 1, (int)((long long)(hx + 1) * Wo / Wi) + 1);
    for (int yy = y0; yy <= y1; ++yy) {
      if (nearest_src(yy, Hi, Ho) != hy) continue;
      for (int xx = x0; xx <= x1; ++xx) {
        if (nearest_src(xx, Wi, Wo) != hx) continue;
        V v = *reinterpret_cast<const V*>(dout + (((size_t)b * Ho + yy) * Wo + xx) * C + cq * EP);
#pragma unroll
        for (int e = 0; e < EP; ++e) a[e] += Elem<T>::to_f(v[e]);
      }
    }
    V o;
#pragma unroll
    for (int e = 0; e < EP; ++e) o[e] = Elem<T>::from_f(a[e]);
    *reinterpret_cast<V*>(dh + (((size_t)b * Hi + hy) * Wi + hx) * C + cq * EP) = o;
  }
}

// src [B,J,N] fp32 (optionally + src2) -> dst [B,N,Jp] T with channels >= J zeroed
// (src2 / dst2 != null: a second tensor of the same shape in the same launch, blockIdx.z = 1 -- the two heads' output gradients, round 6)
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_pad_kernel(const float* __restrict__ src, T* __restrict__ dst, int B, int J, int N,
                                                               int Jp, const float* __restrict__ src2 = nullptr, T* __restrict__ dst2 = nullptr) {
  __shared__ float tile[64][65];
  if (blockIdx.z) { src = src2; dst = dst2; }
  const int b = blockIdx.y, p0 = blockIdx.x * 64;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int j = i / 64, pp = i % 64;
    float v = 0.f;
    if (j < J && p0 + pp < N) v = src[((size_t)b * J + j) * N + p0 + pp];
    if (j < Jp) tile[j][pp] = v;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * Jp; i += 256) {
    const int pp = i / Jp, j = i % Jp;
    if (p0 + pp < N) dst[((size_t)b * N + p0 + pp) * Jp + j] = Elem<T>::from_f(tile[j][pp]);
  }
}

// concat of model.py:208 as an NHWC tensor: channels [0,J) heatmaps, [J,2J) depthmaps, 2J label_img, rest zero.
// pmap/dmap [B,J,N], label [B,1,N] fp32 -> dst [B,N,Cp] T.  grid (N/64, B, ceil(Cp/64)).
template <typename T>
__global__ __launch_bounds__(256) void cat_to_nhwc_kernel(const float* __restrict__ pmap, const float* __restrict__ dmap,
                                                          const float* __restrict__ label, T* __restrict__ dst, int B, int J, int N, int Cp) {
  __shared__ float tile[64][65];
  const int b = blockIdx.y, p0 = blockIdx.x * 64, c0 = blockIdx.z * 64;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int cl = i / 64, pp = i % 64, c = c0 + cl;
    float v = 0.f;
    if (p0 + pp < N) {
      if (c < J) v = pmap[((size_t)b * J + c) * N + p0 + pp];
      else if (c < 2 * J) v = dmap[((size_t)b * J + (c - J)) * N + p0 + pp];
      else if (c == 2 * J) v = label[(size_t)b * N + p0 + pp];
    }
    tile[cl][pp] = v;
  }
  __syncthreads();
  const int cw = min(64, Cp - c0);
  for (int i = threadIdx.x; i < 64 * cw; i += 256) {
    const int pp = i / cw, cl = i % cw;
    if (p0 + pp < N) dst[((size_t)b * N + p0 + pp) * Cp + c0 + cl] = Elem<T>::from_f(tile[cl][pp]);
  }
}
// inverse for the gradients: src [B,N,Cp] T -> gp [B,J,N], gd [B,J,N] fp32 (channels >= 2J dropped)
template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_cat_grad_kernel(const T* __restrict__ src, float* __restrict__ gp, float* __restrict__ gd,
                                                               int B, int J, int N, int Cp) {
  __shared__ float tile[64][65];
  const int b = blockIdx.y, p0 = blockIdx.x * 64, c0 = blockIdx.z * 64;
  const int cw = min(64, Cp - c0);
  for (int i = threadIdx.x; i < 64 * cw; i += 256) {
    const int pp = i / cw, cl = i % cw;
    tile[cl][pp] = (p0 + pp < N) ? Elem<T>::to_f(src[((size_t)b * N + p0 + pp) * Cp + c0 + cl]) : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int cl = i / 64, pp = i % 64, c = c0 + cl;
    if (cl < cw && p0 + pp < N) {
      if (c < J) gp[((size_t)b * J + c) * N + p0 + pp] = tile[cl][pp];
      else if (c < 2 * J) gd[((size_t)b * J + (c - J)) * N + p0 + pp] = tile[cl][pp];
    }
  }
}

// ---- the three layout changes above, vectorised (round 6): 64 pixels x Cp channels per workgroup through an LDS tile whose row pitch (66
// floats) puts the 8-channel reads of the store phase on distinct banks.  Plane side: one float4 of 4 consecutive pixels per access; NHWC
// side: one 16-byte vector (8 bf16 / 4 fp32 channels of a pixel) per access.  The scalar kernels moved 4 and 2 bytes per lane and access
// (cat: 12 us for 24 MB, the gradient's inverse 22 us); same values, same rounding.  N % 64 == 0 (whole tiles), Cp % 8 == 0, Cp <= 128.
struct PlaneSrc { const float* a; const float* b; const float* l; int na, nb; };     // channels [0, na) planes of a, [na, na + nb) of b, na + nb: l (or none)
struct PlaneSrc2 { PlaneSrc s[2]; void* dst[2]; };
template <typename T>
__global__ __launch_bounds__(256) void planes_to_nhwc_kernel(PlaneSrc2 q, int N, int Cp) {
  constexpr int EP = Elem<T>::kPer16B, PITCH = 66;
  typedef typename Vec16<T>::type V;
  __shared__ float tile[128 * PITCH];
  const PlaneSrc s = q.s[blockIdx.z];
  T* __restrict__ dst = reinterpret_cast<T*>(q.dst[blockIdx.z]);
  const int b = blockIdx.y, p0 = blockIdx.x * 64;
  const int nreal = s.na + s.nb + (s.l ? 1 : 0);
  for (int i = threadIdx.x; i < Cp * 16; i += 256) {
    const int c = i >> 4, qd = i & 15;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (c < nreal) {
      const float* pl = c < s.na ? s.a + ((size_t)b * s.na + c) * N : c < s.na + s.nb ? s.b + ((size_t)b * s.nb + (c - s.na)) * N : s.l + (size_t)b * N;
      v = *reinterpret_cast<const f32x4*>(pl + p0 + 4 * qd);
    }
    float* t = tile + c * PITCH + 4 * qd;
    t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
  }
  __syncthreads();
  const int slots = Cp / EP;
  for (int i = threadIdx.x; i < 64 * slots; i += 256) {
    const int pp = i / slots, sl = i - pp * slots;
    V o;
#pragma unroll
    for (int e = 0; e < EP; ++e) o[e] = Elem<T>::from_f(tile[(sl * EP + e) * PITCH + pp]);
    *reinterpret_cast<V*>(dst + ((size_t)b * N + p0 + pp) * Cp + sl * EP) = o;
  }
}
// inverse: src [B,N,Cp] T -> planes (channels beyond na + nb dropped)
struct PlaneDst { float* a; float* b; int na, nb; };
template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_planes_kernel(const T* __restrict__ src, PlaneDst d, int N, int Cp) {
  constexpr int EP = Elem<T>::kPer16B, PITCH = 66;
  typedef typename Vec16<T>::type V;
  __shared__ float tile[128 * PITCH];
  const int b = blockIdx.y, p0 = blockIdx.x * 64;
  const int slots = Cp / EP;
  for (int i = threadIdx.x; i < 64 * slots; i += 256) {
    const int pp = i / slots, sl = i - pp * slots;
    const V v = *reinterpret_cast<const V*>(src + ((size_t)b * N + p0 + pp) * Cp + sl * EP);
#pragma unroll
    for (int e = 0; e < EP; ++e) tile[(sl * EP + e) * PITCH + pp] = Elem<T>::to_f(v[e]);
  }
  __syncthreads();
  const int nreal = d.na + d.nb;
  for (int i = threadIdx.x; i < nreal * 16; i += 256) {
    const int c = i >> 4, qd = i & 15;
    const float* t = tile + c * PITCH + 4 * qd;
    const f32x4 v = {t[0], t[1], t[2], t[3]};
    float* pl = c < d.na ? d.a + ((size_t)b * d.na + c) * N : d.b + ((size_t)b * d.nb + (c - d.na)) * N;
    *reinterpret_cast<f32x4*>(pl + p0 + 4 * qd) = v;
  }
}
static inline bool planes_ok(int N, int Cp, int nreal, const void* p0, const void* p1, const void* p2) {
  static const bool on = PWR_DBG_ENV("PWR_PLANES_VEC", 1) != 0;
  auto al = [](const void* p) { return (reinterpret_cast<size_t>(p) & 15) == 0; };
  return on && N % 64 == 0 && Cp % 8 == 0 && Cp <= 128 && nreal <= Cp && al(p0) && al(p1) && al(p2);
}

template <typename T>
__global__ void axpy_kernel(const T* __restrict__ x, T* __restrict__ y, long long n) {  // y += x
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    y[i] = Elem<T>::from_f(Elem<T>::to_f(y[i]) + Elem<T>::to_f(x[i]));
}

static inline int grid_for(long long total) { long long g = (total + 255) / 256; return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g)); }

}  // namespace pwr

using namespace pwr;

#define PWR_DISPATCH_T(KERNEL, GRID, ...)                                                                       \
  do {                                                                                                          \
    if (dtype == PWR_BF16) hipLaunchKernelGGL((KERNEL<bf16_t>), dim3(GRID), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); \
    else hipLaunchKernelGGL((KERNEL<float>), dim3(GRID), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);      \
  } while (0)

extern "C" int pwr_maxpool_fwd(const void* x, void* y, int B, int H, int W, int C, int dtype, void* stream) {
  const int EP = dtype == PWR_BF16 ? 8 : 4;
  if (C % EP) return PWR_EUNSUPPORTED;
  const int g = grid_for((long long)B * (H / 2) * (W / 2) * (C / EP));
  if (dtype == PWR_BF16) hipLaunchKernelGGL((maxpool_fwd_kernel<bf16_t>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, B, H, W, C);
  else hipLaunchKernelGGL((maxpool_fwd_kernel<float>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, B, H, W, C);
  return (int)hipGetLastError();
}

extern "C" int pwr_maxpool_bwd(const void* x, const void* dh, const void* addend, void* dx, int B, int H, int W, int C, int dtype,
                               void* stream) {
  const int EP = dtype == PWR_BF16 ? 8 : 4;
  if (C % EP) return PWR_EUNSUPPORTED;
  const int g = grid_for((long long)B * ((H + 1) / 2) * ((W + 1) / 2) * (C / EP));
  if (dtype == PWR_BF16) hipLaunchKernelGGL((maxpool_bwd_kernel<bf16_t>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)dh, (const bf16_t*)addend, (bf16_t*)dx, B, H, W, C);
  else hipLaunchKernelGGL((maxpool_bwd_kernel<float>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)x, (const float*)dh, (const float*)addend, (float*)dx, B, H, W, C);
  return (int)hipGetLastError();
}

extern "C" int pwr_upsample_add_fwd(const void* h, const void* skip, void* out, int B, int Hi, int Wi, int Ho, int Wo, int C,
                                    int dtype, void* stream) {
  const int EP = dtype == PWR_BF16 ? 8 : 4;
  if (C % EP) return PWR_EUNSUPPORTED;
  const int g = grid_for((long long)B * Ho * Wo * (C / EP));
  if (dtype == PWR_BF16) hipLaunchKernelGGL((upsample_add_kernel<bf16_t>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)h, (const bf16_t*)skip, (bf16_t*)out, B, Hi, Wi, Ho, Wo, C);
  else hipLaunchKernelGGL((upsample_add_kernel<float>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)h, (const float*)skip, (float*)out, B, Hi, Wi, Ho, Wo, C);
  return (int)hipGetLastError();
}

extern "C" int pwr_upsample_bwd(const void* dout, void* dh, int B, int Hi, int Wi, int Ho, int Wo, int C, int dtype, void* stream) {
  const int EP = dtype == PWR_BF16 ? 8 : 4;
  if (C % EP) return PWR_EUNSUPPORTED;
  const int g = grid_for((long long)B * Hi * Wi * (C / EP));
  if (dtype == PWR_BF16) hipLaunchKernelGGL((upsample_bwd_kernel<bf16_t>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dout, (bf16_t*)dh, B, Hi, Wi, Ho, Wo, C);
  else hipLaunchKernelGGL((upsample_bwd_kernel<float>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)dout, (float*)dh, B, Hi, Wi, Ho, Wo, C);
  return (int)hipGetLastError();
}

extern "C" int pwr_nchw_to_nhwc_pad(const float* src, void* dst, int B, int J, int N, int Jp, int dtype, void* stream) {
  if (Jp > 64 || J > Jp) return PWR_EUNSUPPORTED;
  dim3 grid((N + 63) / 64, B);
  if (planes_ok(N, Jp, J, src, dst, nullptr)) {
    PlaneSrc2 q{};
    q.s[0] = PlaneSrc{src, nullptr, nullptr, J, 0}; q.dst[0] = dst;
    if (dtype == PWR_BF16) hipLaunchKernelGGL((planes_to_nhwc_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, q, N, Jp);
    else hipLaunchKernelGGL((planes_to_nhwc_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, q, N, Jp);
    return (int)hipGetLastError();
  }
  if (dtype == PWR_BF16) hipLaunchKernelGGL((nchw_to_nhwc_pad_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, B, J, N, Jp);
  else hipLaunchKernelGGL((nchw_to_nhwc_pad_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, src, (float*)dst, B, J, N, Jp);
  return (int)hipGetLastError();
}

// two pwr_nchw_to_nhwc_pad of one shape in one launch (the plane and the depth head's output gradients of a stage)
extern "C" int pwr_nchw_to_nhwc_pad_pair(const float* src_a, void* dst_a, const float* src_b, void* dst_b, int B, int J, int N, int Jp, int dtype,
                                         void* stream) {
  if (Jp > 64 || J > Jp) return PWR_EUNSUPPORTED;
  if (!src_a || !dst_a || !src_b || !dst_b) return PWR_EINVAL;
  dim3 grid((N + 63) / 64, B, 2);
  if (planes_ok(N, Jp, J, src_a, dst_a, src_b) && planes_ok(N, Jp, J, dst_b, nullptr, nullptr)) {
    PlaneSrc2 q{};
    q.s[0] = PlaneSrc{src_a, nullptr, nullptr, J, 0}; q.dst[0] = dst_a;
    q.s[1] = PlaneSrc{src_b, nullptr, nullptr, J, 0}; q.dst[1] = dst_b;
    if (dtype == PWR_BF16) hipLaunchKernelGGL((planes_to_nhwc_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, q, N, Jp);
    else hipLaunchKernelGGL((planes_to_nhwc_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, q, N, Jp);
    return (int)hipGetLastError();
  }
  if (dtype == PWR_BF16) hipLaunchKernelGGL((nchw_to_nhwc_pad_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, src_a, (bf16_t*)dst_a, B, J, N, Jp, src_b, (bf16_t*)dst_b);
  else hipLaunchKernelGGL((nchw_to_nhwc_pad_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, src_a, (float*)dst_a, B, J, N, Jp, src_b, (float*)dst_b);
  return (int)hipGetLastError();
}

extern "C" int pwr_add_inplace(const void* x, void* y, long long n, int dtype, void* stream) {
  const int g = grid_for(n);
  if (dtype == PWR_BF16) hipLaunchKernelGGL((axpy_kernel<bf16_t>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, n);
  else hipLaunchKernelGGL((axpy_kernel<float>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, n);
  return (int)hipGetLastError();
}

// ---- bias gradients: column sums ------------------------------------------------------------------
namespace pwr {
// NHWC [M][C] T -> slab[block][C]: 16-byte loads, thread = (row lane, 8/4-channel chunk)
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ x, float* __restrict__ slab, long long M, int C,
                                                             int rows_per_block) {
  constexpr int EP = Elem<T>::kPer16B;
  typedef typename Vec16<T>::type V;
  extern __shared__ float red[];   // [pl][C]
  const int cpp = C / EP, pl = 256 / cpp;
  const int cq = threadIdx.x % cpp, pj = threadIdx.x / cpp;
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  float s[EP];
#pragma unroll
  for (int e = 0; e < EP; ++e) s[e] = 0.f;
  if (pj < pl) {
#pragma unroll 4
    for (long long r = r0 + pj; r < r0 + rows_per_block && r < M; r += pl) {
      V v = *reinterpret_cast<const V*>(x + (size_t)r * C + cq * EP);
#pragma unroll
      for (int e = 0; e < EP; ++e) s[e] += Elem<T>::to_f(v[e]);
    }
#pragma unroll
    for (int e = 0; e < EP; ++e) red[pj * C + cq * EP + e] = s[e];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float t = 0.f;
    for (int j = 0; j < pl; ++j) t += red[j * C + c];
    slab[(size_t)blockIdx.x * C + c] = t;
  }
}
// NCHW fp32 [B][J][N]: part[b][j] = sum_n  (grid (J,B)); then out[j] (+)= sum_b part[b][j]
__global__ __launch_bounds__(256) void planesum_kernel(const float* __restrict__ x, float* __restrict__ part, int B, int J, int N) {
  __shared__ float red[4];
  const int j = blockIdx.x, b = blockIdx.y;
  float s = 0.f;
  for (int i = threadIdx.x; i < N; i += 256) s += x[((size_t)b * J + j) * N + i];
  s = block_sum1(s, red);
  if (threadIdx.x == 0) part[(size_t)b * J + j] = s;
}
}  // namespace pwr

extern "C" int pwr_colsum_blocks(long long M) { long long nb = (M + 255) / 256; return (int)(nb > 128 ? 128 : (nb < 1 ? 1 : nb)); }

extern "C" int pwr_colsum_nhwc(const void* x, float* slab, float* out, long long M, int C, int accumulate, int dtype, void* stream) {
  const int EP = dtype == PWR_BF16 ? 8 : 4;
  if (C % EP || C / EP > 256) return PWR_EUNSUPPORTED;
  const int nb = pwr_colsum_blocks(M);
  const int rpb = (int)((M + nb - 1) / nb);
  const size_t sh = (size_t)(256 / (C / EP)) * C * 4;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == PWR_BF16) hipLaunchKernelGGL((pwr::colsum_partial_kernel<bf16_t>), dim3(nb), dim3(256), sh, s, (const bf16_t*)x, slab, M, C, rpb);
  else hipLaunchKernelGGL((pwr::colsum_partial_kernel<float>), dim3(nb), dim3(256), sh, s, (const float*)x, slab, M, C, rpb);
  return pwr_slab_reduce(slab, out, nb, C, accumulate, stream);
}

extern "C" int pwr_planesum_nchw(const float* x, float* part, float* out, int B, int J, int N, int accumulate, void* stream) {
  hipLaunchKernelGGL(pwr::planesum_kernel, dim3(J, B), dim3(256), 0, (hipStream_t)stream, x, part, B, J, N);
  return pwr_slab_reduce(part, out, B, J, accumulate, stream);
}

extern "C" int pwr_cat_to_nhwc(const float* pmap, const float* dmap, const float* label, void* dst, int B, int J, int N, int Cp, int dtype,
                               void* stream) {
  if (Cp < 2 * J + 1 || Cp % 8) return PWR_EINVAL;
  if (planes_ok(N, Cp, 2 * J + 1, pmap, dmap, label) && planes_ok(N, Cp, 0, dst, nullptr, nullptr)) {
    PlaneSrc2 q{};
    q.s[0] = PlaneSrc{pmap, dmap, label, J, J}; q.dst[0] = dst;
    dim3 g(N / 64, B);
    if (dtype == PWR_BF16) hipLaunchKernelGGL((planes_to_nhwc_kernel<bf16_t>), g, dim3(256), 0, (hipStream_t)stream, q, N, Cp);
    else hipLaunchKernelGGL((planes_to_nhwc_kernel<float>), g, dim3(256), 0, (hipStream_t)stream, q, N, Cp);
    return (int)hipGetLastError();
  }
  dim3 grid((N + 63) / 64, B, (Cp + 63) / 64);
  if (dtype == PWR_BF16) hipLaunchKernelGGL((pwr::cat_to_nhwc_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, pmap, dmap, label, (bf16_t*)dst, B, J, N, Cp);
  else hipLaunchKernelGGL((pwr::cat_to_nhwc_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, pmap, dmap, label, (float*)dst, B, J, N, Cp);
  return (int)hipGetLastError();
}

extern "C" int pwr_nhwc_to_cat_grad(const void* src, float* gp, float* gd, int B, int J, int N, int Cp, int dtype, void* stream) {
  if (Cp < 2 * J + 1 || Cp % 8) return PWR_EINVAL;
  if (planes_ok(N, Cp, 2 * J, src, gp, gd)) {
    const PlaneDst d{gp, gd, J, J};
    dim3 g(N / 64, B);
    if (dtype == PWR_BF16) hipLaunchKernelGGL((nhwc_to_planes_kernel<bf16_t>), g, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, d, N, Cp);
    else hipLaunchKernelGGL((nhwc_to_planes_kernel<float>), g, dim3(256), 0, (hipStream_t)stream, (const float*)src, d, N, Cp);
    return (int)hipGetLastError();
  }
  dim3 grid((N + 63) / 64, B, (Cp + 63) / 64);
  if (dtype == PWR_BF16) hipLaunchKernelGGL((pwr::nhwc_to_cat_grad_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, gp, gd, B, J, N, Cp);
  else hipLaunchKernelGGL((pwr::nhwc_to_cat_grad_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)src, gp, gd, B, J, N, Cp);
  return (int)hipGetLastError();
}
