// Train-step tail of /root/reference/train.py, on the flat buffers of the engine:
//   * the 3-term multi-stage loss of train.py:195-205 and its gradients w.r.t. (heatmaps, depthmaps, uvd)
//       heatmap_loss  = lambda_h * mean_{b,j} sum_pixels (H^ - H)^2
//       depthmap_loss = lambda_d * mean_{b,j} sum_pixels (D^ - D)^2
//       uvd_loss      = mean_{b,j} sum_3 (uvd^ - uvd)^2
//       loss          = sum_stages alpha*uvd_loss + (1-alpha)*(heatmap_loss + depthmap_loss)
//     (with alpha == 1, the reference's default, the dense terms have weight 0: they are skipped, not multiplied by 0)
//   * AdamW / SGD(momentum) on ONE flat fp32 parameter buffer (torch.optim.AdamW / SGD semantics, train.py:139-142).
#include "pwr_common.h"
#include "pwr.h"

namespace pwr {

// partial[block] = sum of squared differences of this block's elements; g = coef * (a - t)   (g may be null)
// (loss != null: a one-block launch finishes the term itself -- loss[0] (+)= scale * its sum, the arithmetic of loss_finish_kernel on one
// partial -- instead of a second launch; the uvd term of train.py:199 is 3 B J numbers)
__global__ __launch_bounds__(256) void sqdiff_grad_kernel(const float* __restrict__ a, const float* __restrict__ t, float* __restrict__ g,
                                                          float coef, float* __restrict__ partial, long long n, float scale = 0.f,
                                                          float* __restrict__ loss = nullptr, int accumulate = 0) {
  __shared__ float red[4];
  float s = 0.f;
  for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long long)gridDim.x * 1024) {
    if (i + 3 < n) {
      const f32x4 av = *reinterpret_cast<const f32x4*>(a + i), tv = *reinterpret_cast<const f32x4*>(t + i);
      const f32x4 d = av - tv;
      s += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
      if (g) *reinterpret_cast<f32x4*>(g + i) = coef * d;
    } else {
      for (long long k = i; k < n; ++k) { const float d = a[k] - t[k]; s += d * d; if (g) g[k] = coef * d; }
    }
  }
  s = block_sum1(s, red);
  if (threadIdx.x == 0) {
    partial[blockIdx.x] = s;
    if (loss) { const float tot = 0.f + s; loss[0] = accumulate ? loss[0] + scale * tot : scale * tot; }
  }
}

// loss[0] (+)= scale * sum_k partial[k]
__global__ void loss_finish_kernel(const float* __restrict__ partial, int n, float scale, float* __restrict__ loss, int accumulate) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float s = 0.f;
  for (int k = 0; k < n; ++k) s += partial[k];
  loss[0] = accumulate ? loss[0] + scale * s : scale * s;
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long long n, float lr, float b1, float b2, float eps, float wd,
                                                    float bc1, float bc2_sqrt, float grad_scale) {
  for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long long)gridDim.x * 1024) {
    if (i + 3 < n) {
      f32x4 pv = *reinterpret_cast<f32x4*>(p + i), gv = grad_scale * *reinterpret_cast<const f32x4*>(g + i);
      f32x4 mv = *reinterpret_cast<f32x4*>(m + i), vv = *reinterpret_cast<f32x4*>(v + i);
      pv = pv * (1.f - lr * wd);
      mv = b1 * mv + (1.f - b1) * gv;
      vv = b2 * vv + (1.f - b2) * (gv * gv);
      f32x4 den;
      den.x = sqrtf(vv.x) / bc2_sqrt + eps; den.y = sqrtf(vv.y) / bc2_sqrt + eps;
      den.z = sqrtf(vv.z) / bc2_sqrt + eps; den.w = sqrtf(vv.w) / bc2_sqrt + eps;
      const float step = lr / bc1;
      pv.x -= step * mv.x / den.x; pv.y -= step * mv.y / den.y; pv.z -= step * mv.z / den.z; pv.w -= step * mv.w / den.w;
      *reinterpret_cast<f32x4*>(p + i) = pv; *reinterpret_cast<f32x4*>(m + i) = mv; *reinterpret_cast<f32x4*>(v + i) = vv;
    } else {
      for (long long k = i; k < n; ++k) {
        const float gg = grad_scale * g[k];
        float pp = p[k] * (1.f - lr * wd);
        const float mm = b1 * m[k] + (1.f - b1) * gg, vv = b2 * v[k] + (1.f - b2) * gg * gg;
        pp -= (lr / bc1) * mm / (sqrtf(vv) / bc2_sqrt + eps);
        p[k] = pp; m[k] = mm; v[k] = vv;
      }
    }
  }
}

// torch.optim.SGD: g += wd*p; buf = first ? g : mu*buf + g; p -= lr*buf
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, long long n,
                                                  float lr, float mu, float wd, int first, float grad_scale) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float gg = grad_scale * g[i] + wd * p[i];
    float b = gg;
    if (mu != 0.f) { b = first ? gg : mu * buf[i] + gg; buf[i] = b; }
    p[i] -= lr * b;
  }
}

}  // namespace pwr

using namespace pwr;

// One loss term: loss[0] (+)= scale * sum (a - t)^2, and g = 2*scale*(a - t) if g != NULL.  partial: >= pwr_loss_blocks(n) floats.
extern "C" int pwr_loss_blocks(long long n) { long long b = (n + 4095) / 4096; return (int)(b > 1024 ? 1024 : (b < 1 ? 1 : b)); }

extern "C" int pwr_loss_sqdiff(const float* a, const float* t, float* g, float scale, float* partial, float* loss, int accumulate,
                               long long n, void* stream) {
  const int nb = pwr_loss_blocks(n);
  hipStream_t s = (hipStream_t)stream;
  if (nb == 1) {
    hipLaunchKernelGGL(sqdiff_grad_kernel, dim3(1), dim3(256), 0, s, a, t, g, 2.f * scale, partial, n, scale, loss, accumulate);
    return (int)hipGetLastError();
  }
  hipLaunchKernelGGL(sqdiff_grad_kernel, dim3(nb), dim3(256), 0, s, a, t, g, 2.f * scale, partial, n, 0.f, (float*)nullptr, 0);
  hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(64), 0, s, partial, nb, scale, loss, accumulate);
  return (int)hipGetLastError();
}

// torch.optim.AdamW step `step` (1-based) on flat buffers; grad_scale multiplies the gradient first (1/world size, or the
// inverse AMP loss scale).
extern "C" int pwr_adamw_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                              float weight_decay, int step, float grad_scale, void* stream) {
  const float bc1 = 1.f - powf(beta1, (float)step), bc2s = sqrtf(1.f - powf(beta2, (float)step));
  long long nb = (n + 1023) / 1024;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(adamw_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps, weight_decay,
                     bc1, bc2s, grad_scale);
  return (int)hipGetLastError();
}

extern "C" int pwr_sgd_step(float* p, const float* g, float* buf, long long n, float lr, float momentum, float weight_decay, int first,
                            float grad_scale, void* stream) {
  long long nb = (n + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(sgd_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, p, g, buf, n, lr, momentum, weight_decay, first,
                     grad_scale);
  return (int)hipGetLastError();
}
