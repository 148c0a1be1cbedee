// Input pipeline on the device: what /root/reference/datasets.py:243-299 does per sample on the CPU with
// numpy + OpenCV -- crop a cube around the hand's centre of mass out of the raw depth frame, cut the depth range, centre on the
// COM, cv2.resize to S x S, optional rotation / scale augmentation (cv2.warpAffine), cv2.resize to the P x P label image, mask,
// normalisation by the cube size -- as three gather kernels over a batch of raw frames that are already in HBM.
//
// OpenCV semantics restated (oracle/preprocess_ref.py has the CPU twin and the citations):
//   resize, INTER_LINEAR, float32: fx = float((dx + 0.5) * scale - 0.5), index clamped to the source, horizontal taps first
//   warpAffine, INTER_LINEAR, BORDER_CONSTANT 0: inverse map in fixed point (AB_BITS 10, 1/32-pixel weights)
// All arithmetic orders follow the numpy / OpenCV float32 evaluation (no FMA contraction: the library is built with
// -ffp-contract=off), so the outputs are bit-comparable with the CPU restatement.
#include "pwr_common.h"
#include "pwr.h"

namespace pwr {

struct Tap { int i0, i1; float a; };
__device__ __forceinline__ Tap resize_tap(int d, int n_src, double scale) {
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (s < 0) { s = 0; f = 0.f; }
  if (s >= n_src - 1) { s = n_src - 1; f = 0.f; }
  Tap t; t.i0 = s; t.i1 = min(s + 1, n_src - 1); t.a = f;
  return t;
}

// crop window of sample b: rows r0 + [0, side), cols c0 + [0, side) of the raw frame (zero outside the frame); a depth value is kept
// iff com_z - cube < v < com_z + cube (compared in double like numpy does for a float32 array against a float64 scalar) and then
// centred: float(double(v) - com_z)                                                            datasets.py:248-252
__device__ __forceinline__ float crop_value(const float* __restrict__ img, int H, int W, int r, int c, double cz, double cube) {
  if (r < 0 || r >= H || c < 0 || c >= W) return 0.f;
  const float v = img[(size_t)r * W + c];
  const double dv = (double)v;
  if (!(dv > cz - cube && dv < cz + cube)) return 0.f;
  return v > 0.f ? (float)(dv - cz) : v;
}

// geo[b] = {r0, c0, side}; comz[b], cube[b] doubles.  out [B,S,S] (millimetres relative to the COM, 0 = background)
__global__ __launch_bounds__(256) void crop_resize_kernel(const float* __restrict__ depth, const int* __restrict__ geo,
                                                          const double* __restrict__ comz, const double* __restrict__ cube,
                                                          float* __restrict__ out, int H, int W, int S) {
  const int b = blockIdx.y;
  const int r0 = geo[b * 3 + 0], c0 = geo[b * 3 + 1], side = geo[b * 3 + 2];
  const double scale = (double)side / (double)S;
  const double cz = comz[b], cb = cube[b];
  const float* __restrict__ img = depth + (size_t)b * H * W;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < S * S; i += gridDim.x * 256) {
    const int y = i / S, x = i - y * S;
    const Tap tx = resize_tap(x, side, scale), ty = resize_tap(y, side, scale);
    const float a00 = crop_value(img, H, W, r0 + ty.i0, c0 + tx.i0, cz, cb), a01 = crop_value(img, H, W, r0 + ty.i0, c0 + tx.i1, cz, cb);
    const float a10 = crop_value(img, H, W, r0 + ty.i1, c0 + tx.i0, cz, cb), a11 = crop_value(img, H, W, r0 + ty.i1, c0 + tx.i1, cz, cb);
    const float w0 = 1.f - tx.a, h0 = 1.f - ty.a;
    const float top = a00 * w0 + a01 * tx.a, bot = a10 * w0 + a11 * tx.a;       // HResize
    out[(size_t)b * S * S + i] = top * h0 + bot * ty.a;                         // VResize
  }
}

// dst = warpAffine(src, M) * scale.  minv[b] = the six doubles of the INVERTED matrix (dst -> src), scale[b] float.
__global__ __launch_bounds__(256) void warp_affine_kernel(const float* __restrict__ src, const double* __restrict__ minv,
                                                          const float* __restrict__ scale, float* __restrict__ dst, int S) {
  const int b = blockIdx.y;
  const double m00 = minv[b * 6 + 0], m01 = minv[b * 6 + 1], m02 = minv[b * 6 + 2];
  const double m10 = minv[b * 6 + 3], m11 = minv[b * 6 + 4], m12 = minv[b * 6 + 5];
  const float sc = scale[b];
  const float* __restrict__ s = src + (size_t)b * S * S;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < S * S; i += gridDim.x * 256) {
    const int y = i / S, x = i - y * S;
    const long long adelta = (long long)rint(m00 * (double)x * 1024.0), bdelta = (long long)rint(m10 * (double)x * 1024.0);
    const long long X0 = (long long)rint((m01 * (double)y + m02) * 1024.0) + 16, Y0 = (long long)rint((m11 * (double)y + m12) * 1024.0) + 16;
    const long long X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
    const int ix = (int)(X >> 5), iy = (int)(Y >> 5);
    const float fx = (float)(X & 31) / 32.f, fy = (float)(Y & 31) / 32.f;
    float v = 0.f;
    if (ix >= -1 && ix < S && iy >= -1 && iy < S) {
      auto at = [&](int r, int c) { return (r >= 0 && r < S && c >= 0 && c < S) ? s[(size_t)r * S + c] : 0.f; };
      const float w00 = (1.f - fy) * (1.f - fx), w01 = (1.f - fy) * fx, w10 = fy * (1.f - fx), w11 = fy * fx;
      v = at(iy, ix) * w00 + at(iy, ix + 1) * w01 + at(iy + 1, ix) * w10 + at(iy + 1, ix + 1) * w11;
    }
    dst[(size_t)b * S * S + i] = v * sc;
  }
}

// label = resize(img, P x P) (millimetres); mask = label != 0; img_n = img / cube; label_n = label / cube     datasets.py:297-299, 378-380
__global__ __launch_bounds__(256) void label_mask_norm_kernel(const float* __restrict__ img, const float* __restrict__ cube_f,
                                                              float* __restrict__ img_n, float* __restrict__ label_n,
                                                              float* __restrict__ mask, int S, int P) {
  const int b = blockIdx.y;
  const double scale = (double)S / (double)P;
  const float cb = cube_f[b];
  const float* __restrict__ s = img + (size_t)b * S * S;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < S * S; i += gridDim.x * 256) {
    img_n[(size_t)b * S * S + i] = s[i] / cb;
    if (i < P * P) {
      const int y = i / P, x = i - y * P;
      const Tap tx = resize_tap(x, S, scale), ty = resize_tap(y, S, scale);
      const float w0 = 1.f - tx.a, h0 = 1.f - ty.a;
      const float top = s[(size_t)ty.i0 * S + tx.i0] * w0 + s[(size_t)ty.i0 * S + tx.i1] * tx.a;
      const float bot = s[(size_t)ty.i1 * S + tx.i0] * w0 + s[(size_t)ty.i1 * S + tx.i1] * tx.a;
      const float l = top * h0 + bot * ty.a;
      label_n[(size_t)b * P * P + i] = l / cb;
      mask[(size_t)b * P * P + i] = l != 0.f ? 1.f : 0.f;
    }
  }
}

}  // namespace pwr

using namespace pwr;

extern "C" int pwr_crop_resize(const float* depth, const int* geo, const double* com_z, const double* cube, float* out, int B, int H, int W,
                               int S, void* stream) {
  if (B <= 0 || H <= 0 || W <= 0 || S <= 1) return PWR_EINVAL;
  hipLaunchKernelGGL(crop_resize_kernel, dim3((S * S + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, depth, geo, com_z, cube, out, H, W, S);
  return (int)hipGetLastError();
}

extern "C" int pwr_warp_affine(const float* src, const double* minv, const float* scale, float* dst, int B, int S, void* stream) {
  if (B <= 0 || S <= 1 || src == dst) return PWR_EINVAL;
  hipLaunchKernelGGL(warp_affine_kernel, dim3((S * S + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, src, minv, scale, dst, S);
  return (int)hipGetLastError();
}

extern "C" int pwr_label_mask_normalize(const float* img, const float* cube, float* img_n, float* label_n, float* mask, int B, int S, int P,
                                        void* stream) {
  if (B <= 0 || S <= 1 || P <= 1 || P > S) return PWR_EINVAL;
  hipLaunchKernelGGL(label_mask_norm_kernel, dim3((S * S + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, img, cube, img_n, label_n, mask, S, P);
  return (int)hipGetLastError();
}
