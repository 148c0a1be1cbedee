// Dense training targets on the device (/root/reference/datasets.py:285-294, :365-383).
//
// The reference builds, per sample and joint, on the CPU: a bilinear 2x2 splat of the joint position in label pixels
// (utils.py:37-61), cv2.GaussianBlur(k x k, sigma) of it (utils.py:63-64), and the depth-offset map
// (d_j - label_image) * (heatmap_j > 0) * mask (datasets.py:369-375), then ships [B,J,P,P] x 2 fp32 to the GPU.  Here they
// are generated from the [B,J,3] normalised joints where they are consumed: one workgroup per (b, j) map, the blur evaluated
// directly from the four splat weights (separable Gaussian, BORDER_REFLECT_101 like OpenCV's default), HBM traffic = the two
// output maps + label_img / mask.
#include "pwr_common.h"
#include "pwr.h"

namespace pwr {

__device__ __forceinline__ int reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
  return i;
}

#define PWR_MAXK 15
__global__ __launch_bounds__(256) void make_targets_kernel(const float* __restrict__ uvd, const float* __restrict__ label,
                                                           const float* __restrict__ mask, float* __restrict__ heat,
                                                           float* __restrict__ dmap, int J, int P, int ksize, float sigma) {
  const int bj = blockIdx.x, b = bj / J;
  __shared__ float g[PWR_MAXK];
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < ksize; ++i) { const float t = (float)i - 0.5f * (float)(ksize - 1); g[i] = expf(-(t * t) / (2.f * sigma * sigma)); s += g[i]; }
    for (int i = 0; i < ksize; ++i) g[i] /= s;
  }
  __syncthreads();
  const float un = uvd[(size_t)bj * 3 + 0], vn = uvd[(size_t)bj * 3 + 1], dn = uvd[(size_t)bj * 3 + 2];
  // position and splat weights in double like the reference's numpy code (the support `heat > 0` hinges on du, dv == 0 exactly)
  const double u = (double)un * (double)(P - 1) + (double)(P / 2), v = (double)vn * (double)(P - 1) + (double)(P / 2);   // datasets.py:287-289
  const double fu = floor(u), fv = floor(v);
  const int lu = (int)fu, lv = (int)fv;
  const double du = u - fu, dv = v - fv;
  const double mind = fmax(du + dv - 1.0, 0.0), maxd = fmin(du, dv);
  const double wdd = 0.5 * (maxd + mind);                                                            // utils.py:46-53
  const float wd = (float)wdd, wb = (float)(du - wdd), wc = (float)(dv - wdd), wa = (float)(1.0 + wdd - du - dv);
  // utils.py:54-57 index the map with numpy's rule: -P <= i < P, negative indices WRAP (a joint up to P pixels left of / above the map
  // lands on the opposite border: lu = -1 -> columns P-1 and 0); the reference fails only for lu + 1 >= P, lu < -P or a NaN position
  const bool inside = u == u && v == v && fu >= -(double)P && fv >= -(double)P && fu + 1.0 < (double)P && fv + 1.0 < (double)P;
  const int cu0 = inside ? (lu + P) % P : 0, cu1 = inside ? (lu + 1 + P) % P : 0;
  const int rv0 = inside ? (lv + P) % P : 0, rv1 = inside ? (lv + 1 + P) % P : 0;
  const int r = ksize / 2;
  float* __restrict__ ho = heat + (size_t)bj * P * P;
  float* __restrict__ dout = dmap + (size_t)bj * P * P;
  const float* __restrict__ lb = label + (size_t)b * P * P;
  const float* __restrict__ mk = mask + (size_t)b * P * P;
  for (int i = threadIdx.x; i < P * P; i += 256) {
    const int y = i / P, x = i - y * P;
    float h = 0.f;
    if (inside) {
      // separable blur of the 2x2 splat: column weights for source columns lu, lu+1 and row weights for lv, lv+1 (wrapped)
      float cx0 = 0.f, cx1 = 0.f, ry0 = 0.f, ry1 = 0.f;
      for (int t = 0; t < ksize; ++t) {
        const int sx = reflect101(x + t - r, P), sy = reflect101(y + t - r, P);
        if (sx == cu0) cx0 += g[t];
        if (sx == cu1) cx1 += g[t];
        if (sy == rv0) ry0 += g[t];
        if (sy == rv1) ry1 += g[t];
      }
      h = ry0 * (wa * cx0 + wb * cx1) + ry1 * (wc * cx0 + wd * cx1);
    }
    ho[i] = h;
    dout[i] = h > 0.f ? (dn - lb[i]) * mk[i] : 0.f;
  }
}

}  // namespace pwr

// uvd: [B,J,3] normalised joints (datasets.py:382-384); label_img, mask: [B,P,P]; heatmaps, depthmaps: [B,J,P,P] (train.py:197-198
// targets).  A joint for which utils.generate_heatmap raises (footprint beyond the right / bottom border, more than P pixels left / above, or
// NaN) gets all-zero maps; preprocess_batch reports such samples (the reference falls back to the un-augmented sample or drops it).
extern "C" int pwr_make_targets(const float* uvd, const float* label_img, const float* mask, float* heatmaps, float* depthmaps, int B,
                                int J, int P, int ksize, float sigma, void* stream) {
  if (ksize < 1 || ksize > PWR_MAXK || !(ksize & 1) || !(sigma > 0.f)) return PWR_EINVAL;
  hipLaunchKernelGGL(pwr::make_targets_kernel, dim3(B * J), dim3(256), 0, (hipStream_t)stream, uvd, label_img, mask, heatmaps, depthmaps, J, P,
                     ksize, sigma);
  return (int)hipGetLastError();
}
