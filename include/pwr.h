/* pwr.h -- C ABI of libpwr_hip.so: the MI355X (gfx950) kernels behind PixelwiseRegression.forward.
 *
 * The reference (IcarusWizard/PixelwiseRegression) is pure Python: its "plugin API" for this path is
 * `from model import PixelwiseRegression` (train.py:9, test.py:8).  There is no FFI in the reference,
 * so the boundary below is what a maintainer binds (ctypes) to replace the ATen op
 * sequences of model.py with one call each.  Each entry point cites the reference lines it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless it says "host"; plain C types only, no torch types;
 *   - activations inside the network are NHWC ("pixel-major, channel-minor") in dtype PWR_F32 or
 *     PWR_BF16; tensors that cross the nn.Module boundary (img, label_img, mask, heatmaps, depthmaps,
 *     uvd, parameters, gradients) are fp32 in the reference's layouts (NCHW / OIHW);
 *   - calls are asynchronous on `stream` (a hipStream_t passed as void*), allocate nothing and never
 *     synchronise; workspaces are caller-provided;
 *   - return value: 0 = ok, >0 = hipError_t, <0 = bad argument (PWR_E*).
 */
#ifndef PWR_H_
#define PWR_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PWR_F32 0
#define PWR_BF16 1

#define PWR_EINVAL (-1)
#define PWR_EUNSUPPORTED (-2)
#define PWR_ENOSPACE (-3)

#define PWR_HEATMAP_SOFTMAX 0 /* model.py:81-85 */
#define PWR_HEATMAP_SUM 1     /* model.py:86-90 */

/* ABI version of this header; pwr_abi_version() must return the same number. */
#define PWR_ABI_VERSION 1
int pwr_abi_version(void);

/* ---------------------------------------------------------------------------------------------
 * Decoder (SURVEY.md section 8 a-D)
 * ------------------------------------------------------------------------------------------- */

/* Replaces PlaneRegression.forward after its conv head (model.py:79-97) and DepthRegression.forward
 * after its conv head (model.py:123-132):
 *   p = softmax_i(w[j]*z) | (relu(z)+1e-14)/sum;  u,v = sum p*grid (utils.py:24-35);
 *   d = sum (p*m)*(m*(D+L)) / (sum p*m + 1e-14).
 * z, D: [B,J,P,P]; L, m: [B,1,P,P]; w: [J] (ignored for PWR_HEATMAP_SUM); p_out: [B,J,P,P]; uvd_out: [B,J,3]. */
int pwr_decode_fwd(const float* z, const float* D, const float* L, const float* m, const float* w, float* p_out,
                   float* uvd_out, int B, int J, int P, int method, void* stream);

/* Backward of the above (what autograd derives for model.py:79-97,123-132).
 * p, uvd: outputs of pwr_decode_fwd.  gH, gD_in: upstream gradients of heatmaps / depthmaps, [B,J,P,P],
 * either may be NULL (= zeros).  gU: [B,J,3].  Outputs gz_out, gD_out: [B,J,P,P] (gD_out may alias gD_in);
 * gw_part: [B*J] per-map partial of dL/dw (reduce with pwr_decode_gw_reduce), may be NULL. */
int pwr_decode_bwd(const float* p, const float* z, const float* D, const float* L, const float* m, const float* w,
                   const float* uvd, const float* gH, const float* gD_in, const float* gU, float* gz_out,
                   float* gD_out, float* gw_part, int B, int J, int P, int method, void* stream);

/* gw[j] (+)= sum_b gw_part[b*J+j], fixed summation order. */
int pwr_decode_gw_reduce(const float* gw_part, float* gw, int B, int J, int accumulate, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PWR_H_ */
