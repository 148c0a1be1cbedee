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
#define PWR_ABI_VERSION 7   /* 7 (round 6): pwr_conv_dgrad_fold_stats_pair; 6 (round 6): pwr_norm_bwd_apply_from_partial, pwr_norm_bwd_params_from_partial, pwr_norm_bwd_params_group, pwr_norm_finalize_partial_pair, pwr_engine_pack_beside_forward, pwr_norm_apply, pwr_norm_stats_fused_src, pwr_nchw_to_nhwc_pad_pair; 5 (round 5): pwr_conv_fwd_nchw_pair, the order field of the pack records and the tag bit of a fragment-order pack; 4 (round 4): pwr_conv_dgrad_stats_pair, pwr_conv_wgrad_pair, pwr_engine_wait_segment, pwr_resblock_bwd_small_x, pwr_norm_bwd_from_partial_pair, pwr_conv_fwd_stats mode 1; 3 (round 3): experiment entry points removed, debugging aids moved to pwr_debug.h */
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


/* ---------------------------------------------------------------------------------------------
 * Convolutions on the matrix cores (SURVEY.md section 8 a-C): every torch.nn.Conv2d of model.py with
 * Cin % 4 == 0 -- stem :171-185, stage input :137 (stage 0), ResBlock :13-19, heads :54-65 / :103-114.
 * Activations NHWC in `dtype`; weights come from the flat fp32 parameter buffer re-packed by
 * pwr_pack_weights (one launch for all layers).
 * ------------------------------------------------------------------------------------------- */

/* Output channels are padded to the N tile of the kernel (32 / 64 / 128). */
int pwr_conv_out_pad(int cout);

/* Bytes of one packed weight tensor. kind: 0 forward, 1 data-gradient of a stride-1 conv (taps flipped,
 * in/out swapped), 2 data-gradient of a stride-2 conv (gather form). */
size_t pwr_conv_pack_bytes(int cout, int cin, int ksize, int kind, int dtype);

/* descs_dev: device array of n_desc records {int64 src_off (floats into flat_params), int64 dst_off (bytes
 * into packs), int32 Cout, Cin, ksize, kind, rows_pad, KCH, dtype, pad}.  rows_pad = pwr_conv_out_pad(rows),
 * KCH = ceil(kdim / (dtype==PWR_BF16 ? 32 : 16)) with rows/kdim = (Cout,Cin) for kind 0, (Cin,Cout) else.
 * The last int32 of a record (`pad` until round 5) is the pack's ORDER: 0 = the standard [tap][kch][row][32 bf16] image every conv
 * kernel reads; 1 = the fragment order of the weight-stationary 128 -> 128 3x3 bf16 conv ([tap][kch][32-row group][k half][lane][8 bf16],
 * kind 0, bf16, Cout = Cin = 128, ksize 3 only -- same bytes, another permutation).  A fragment-order pack is handed to pwr_conv_fwd /
 * pwr_conv_fwd_stats / pwr_conv_fwd_pair with BIT 0 OF ITS ADDRESS SET (packs are 256-byte aligned, the bit is free): the launch
 * strips it, takes the weight-stationary kernel and returns PWR_EINVAL if the shape is not one that kernel takes (H % 4, W % 32,
 * stride 1, NHWC output, no residual, >= 16 tiles) -- it never reads a pack in the wrong order. */
int pwr_pack_weights(const float* flat_params, void* packs, const void* descs_dev, int n_desc, void* stream);

/* y = conv(NR(x)) + bias (+ residual).  NR(x) = (x - mean[b,c]) * scale[b,c] + beta[b,c], then ReLU if relu_in;
 * skipped when in_norm == NULL: the InstanceNorm/BatchNorm + ReLU that precedes the conv in model.py, fused
 * into the operand load.  in_norm is the [4][B][Cin] state written by pwr_norm_stats.  x: [B,H,W,Cin]; y: [B,Ho,Wo,Cout] (NULL to skip); y_nchw: fp32 [B,Cout,Ho,Wo] (NULL to
 * skip; used by the heads' last conv, model.py:64/:113).  ksize in {1,3,5,7} (the scripts' --filter_size, train.py:47), pad = ksize/2, stride in {1,2}.
 * mode 0: convolution.  mode 1: data-gradient of a stride-2 conv: x is dy [B,H,W,Cin], y is [B,2H,2W,Cout],
 * wpack of kind 2.  The data-gradient of a stride-1 conv is mode 0 with a kind-1 pack. */
int pwr_conv_fwd(const void* x, const void* wpack, const float* bias, const float* in_norm, int relu_in,
                 const void* residual, void* y, float* y_nchw, int B, int H, int W, int Cin, int Cout, int ksize, int stride,
                 int mode, int dtype, void* stream);

/* pwr_conv_fwd that also writes per-channel column statistics of its output from the epilogue, so that the norm which
 * follows the conv in model.py (forward: st_partial), or the norm backward that consumes a data gradient (nb_partial: the
 * launch computes g = dL/d relu(norm(nb_y)); nb_state is that norm's [4][B][Cout] state), needs no reduction pass of its own.
 * Exactly one of st_partial ([B * chunks][3][Cout] fp32: shifted sum, shifted sum of squares, the tile's shift) /
 * nb_partial ([B * chunks][2][Cout]); chunks = pwr_conv_stats_chunks(...) > 0 (0: shape not supported -- use pwr_conv_fwd and
 * pwr_norm_stats / pwr_norm_bwd).  mode 1 (the data gradient of a stride-2 3x3 conv, model.py:182, H x W = the gradient's map): nb_partial
 * only; the four parity classes (one launch) write one slab of 4 x (H / 4) x (W / 32) rows per sample. */
int pwr_conv_stats_chunks(int H, int W, int Cin, int Cout, int ksize, int stride, int mode, int dtype);
int pwr_conv_fwd_stats(const void* x, const void* wpack, const float* bias, const float* in_norm, int relu_in,
                       const void* residual, void* y, int B, int H, int W, int Cin, int Cout, int ksize, int stride, int mode,
                       float* st_partial, const void* nb_y, const float* nb_state, float* nb_partial, int nb_relu, int dtype,
                       void* stream);

/* Two pwr_conv_fwd_stats calls (forward statistics form: st_partial, stride 1, no residual) of ONE shape on different tensors / weights
 * in ONE launch -- the two regression heads of a stage (model.py:54-65 / :103-114).  PWR_EUNSUPPORTED: no pair kernel for the shape, launch
 * them one after the other.  Results are those of the two single launches, bit for bit. */
int pwr_conv_fwd_stats_pair(const void* xa, const void* wa, const float* bias_a, const float* in_norm_a, void* ya, float* st_partial_a,
                            const void* xb, const void* wb, const float* bias_b, const float* in_norm_b, void* yb, float* st_partial_b,
                            int relu_in, int B, int H, int W, int Cin, int Cout, int ksize, int dtype, void* stream);

/* Two data gradients of stride-1 3x3 convs of ONE shape in ONE launch -- pwr_conv_fwd_stats in its norm-backward form (x = dy, kind-1 pack,
 * no bias, no prologue, nb_partial) for both regression heads of a stage, which walk their convs backwards in lock-step
 * (autograd of model.py:54-65 / :103-114).  PWR_EUNSUPPORTED: no pair kernel for the shape.  Results are those of the two single launches, bit for bit. */
/* Two pwr_conv_fwd launches that write fp32 NCHW maps only (y == NULL), of ONE shape, as one launch: the two regression heads' last convs
 * (model.py:64 and :113, 128 -> J).  PWR_EUNSUPPORTED when the shape has no such launch (bf16, 3x3, stride 1, Cin = 128, Cout <= 32,
 * H % 4 == 0, W % 32 == 0): call pwr_conv_fwd twice.  Results are those of the two single launches, bit for bit. */
int pwr_conv_fwd_nchw_pair(const void* xa, const void* wa, const float* bias_a, const float* in_norm_a, float* ya_nchw,
                           const void* xb, const void* wb, const float* bias_b, const float* in_norm_b, float* yb_nchw,
                           int relu_in, int B, int H, int W, int Cin, int Cout, int ksize, int dtype, void* stream);

int pwr_conv_dgrad_stats_pair(const void* dya, const void* wa, void* dxa, const void* nb_y_a, const float* nb_state_a, float* nb_partial_a,
                              const void* dyb, const void* wb, void* dxb, const void* nb_y_b, const float* nb_state_b, float* nb_partial_b,
                              int nb_relu, int B, int H, int W, int Cin, int Cout, int ksize, int dtype, void* stream);

/* pwr_conv_dgrad_stats_pair with the norm backward of the layer ABOVE folded into its staging (round 6): g_* are the RAW gradients
 * dL/d relu(norm(fb_y_*)) that the data gradient above wrote, fb_partial_* its norm-backward sums (fb_pchunks slab rows of [2][Cin] per
 * sample), fb_state_* the norm's state [4][B][Cin].  The launch computes dy = pwr_norm_bwd_apply_from_partial's result on the way into LDS
 * (same bits), convolves it, and writes it to fb_dy_* (the weight gradients' operand; must not alias g_*).  bf16, 3x3, Cin = Cout = 128,
 * H % 4 == 0, W % 32 == 0; PWR_EUNSUPPORTED otherwise: call pwr_norm_bwd_apply_from_partial + pwr_conv_dgrad_stats_pair. */
int pwr_conv_dgrad_fold_stats_pair(const void* ga, const void* wa, void* dxa, const void* nb_y_a, const float* nb_state_a, float* nb_partial_a,
                                   const void* fb_y_a, const float* fb_state_a, const float* fb_partial_a, void* fb_dy_a,
                                   const void* gb, const void* wb, void* dxb, const void* nb_y_b, const float* nb_state_b, float* nb_partial_b,
                                   const void* fb_y_b, const float* fb_state_b, const float* fb_partial_b, void* fb_dy_b,
                                   int fb_pchunks, int fb_relu, int nb_relu, int B, int H, int W, int Cin, int Cout, int ksize, int dtype,
                                   void* stream);

size_t pwr_conv_wgrad_slab_bytes(int cout, int cin, int ksize, int splits);
/* dw[cout_real][cin_real][k][k] (+)= sum_{b,pixels} dy * NR(x)  (OIHW fp32, the layout of the nn.Parameter gradient).
 * x: forward input [B,H,W,Cin]; dy: [B,Ho,Wo,Cout]; Cin / Cout may include zero-padded channels beyond cin_real /
 * cout_real (the stage-input concat is padded 2J+1 -> multiple of 8, the heads' dy J -> multiple of 8);
 * slab: workspace of pwr_conv_wgrad_slab_bytes; `splits` partitions the pixel (K) dimension over workgroups. */
int pwr_conv_wgrad(const void* x, const void* dy, const float* in_norm, int relu_in, float* slab,
                   float* dw, int accumulate, int B, int H, int W, int Cin, int cin_real, int Cout, int cout_real, int ksize,
                   int stride, int splits, int dtype, void* stream);

/* Two pwr_conv_wgrad calls of ONE geometry (3x3, stride 1, bf16, Cin and Cout multiples of 128, no padded channels) in ONE kernel launch
 * + ONE reduce launch: the weight gradients of the two regression heads' convs of the same depth (model.py:55-63 / :104-112).
 * slab: 2 x pwr_conv_wgrad_slab_bytes(Cout, Cin, 3, splits).  in_norm_a / in_norm_b: both NULL or both set.  PWR_EUNSUPPORTED: shape
 * without a pair kernel.  Bit-identical to two pwr_conv_wgrad calls with the same `splits`. */
int pwr_conv_wgrad_pair(const void* xa, const void* dya, const float* in_norm_a, float* dwa, const void* xb, const void* dyb,
                        const float* in_norm_b, float* dwb, int relu_in, float* slab, int B, int H, int W, int Cin, int Cout, int splits,
                        int dtype, void* stream);

/* Stem conv with Cin = 1 (model.py:165).  img: fp32 [B,S,S]; w: OIHW fp32 [C0,1,k,k]; y: [B,S,S,C0]. */
int pwr_stem_conv_fwd(const float* img, const float* w, const float* bias, void* y, int B, int S, int C0, int ksize,
                      int dtype, void* stream);
int pwr_stem_conv_wgrad_blocks(int B, int S); /* slab floats = blocks * C0 * k * k */
int pwr_stem_conv_wgrad(const float* img, const void* dy, float* slab, float* dw, int accumulate, int B, int S, int C0,
                        int ksize, int dtype, void* stream);
int pwr_slab_reduce(const float* slab, float* out, int S, int n, int accumulate, void* stream);

/* Stage-input 1x1 conv of stages >= 1 (model.py:137) fused with the concat of model.py:208: reads
 * heatmaps [B,J,N], depthmaps [B,J,N], label_img [B,1,N] (fp32 NCHW, N = P*P), writes y [B,N,F].  w: [F,2J+1]. */
int pwr_catconv_fwd(const float* pmap, const float* dmap, const float* label, const float* w, const float* bias, void* y,
                    int B, int N, int J, int F, int dtype, void* stream);
/* gradients w.r.t. heatmaps (gp) and depthmaps (gd), fp32 [B,J,N], overwritten */
int pwr_catconv_dgrad(const void* dy, const float* w, float* gp, float* gd, int B, int N, int J, int F, int dtype,
                      void* stream);
int pwr_catconv_wgrad_blocks(int B, int N); /* slab floats = blocks * (2J+2) * F */
int pwr_catconv_wgrad(const float* pmap, const float* dmap, const float* label, const void* dy, float* slab, float* dw,
                      float* db, int accumulate, int B, int N, int J, int F, int dtype, void* stream);

/* The weight gradients of SEVERAL layers in one launch (+ two split-K reduce launches): the 24 conv layers per stage on the 16x16 ..
 * 2x2 maps of the inner hourglass (model.py:25-47 below the 32x32 level) are microseconds of arithmetic each.  bf16, stride 1,
 * ksize 1 or 3, at most 48 jobs; every job as for pwr_conv_wgrad with relu_in / in_norm per job, accumulate = 0.  slab: workspace
 * of pwr_conv_wgrad_group_slab_bytes (0 = unsupported job list). */
typedef struct {
  const void* x; const void* dy; const float* in_norm; float* dw;
  int H, W, Cin, cin_real, Cout, cout_real, ksize, relu_in;
} pwr_wgrad_job;
size_t pwr_conv_wgrad_group_slab_bytes(const pwr_wgrad_job* jobs, int njobs, int B);
int pwr_conv_wgrad_group(const pwr_wgrad_job* jobs, int njobs, float* slab, int B, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Norm + ReLU (model.py: every `norm(...)`, ReLU pair).  mode: 0 InstanceNorm2d, 1 BatchNorm2d training,
 * 2 BatchNorm2d eval (running statistics).  Output `state` is [4][B][C] fp32 = mean, rstd, scale = gamma*rstd, beta
 * (consumed by pwr_conv_fwd / pwr_conv_wgrad / pwr_norm_bwd).
 * ------------------------------------------------------------------------------------------- */
int pwr_norm_chunks(int B, int HW);
size_t pwr_norm_partial_bytes(int B, int HW, int C);
int pwr_norm_stats(const void* y, const float* gamma, const float* beta, float* running_mean, float* running_var,
                   float* partial, float* state, int B, int HW, int C, int mode, float eps, float momentum, int dtype,
                   void* stream);
/* Round 6: the tensor's PRODUCER fused into the statistics launch (InstanceNorm): src 1: y = maxpool2x2(xa [B,2H,2W,C]) (model.py:40);
 * src 2: y = xa [B,H,W,C] + nearest-upsample(xh [B,H/2,W/2,C]) (model.py:45-47).  y is written; y and `state` are bit-identical to
 * pwr_maxpool_fwd / pwr_upsample_add_fwd followed by pwr_norm_stats(mode 0).  PWR_EUNSUPPORTED for H W <= 512 (pwr_norm_stats' one-block
 * form): issue the separate launches then.  partial: pwr_norm_partial_bytes(B, H W, C). */
int pwr_norm_stats_fused_src(int src, const void* xa, const void* xh, void* y, const float* gamma, const float* beta, float* partial, float* state,
                             int B, int H, int W, int C, float eps, int dtype, void* stream);
/* state from the per-tile sums of pwr_conv_fwd_stats (st_partial); mode 0 / 1. */
int pwr_norm_finalize_partial(const float* partial, int chunks, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, float* state, int B, int HW, int C, int mode, float eps,
                              float momentum, void* stream);
/* two pwr_norm_finalize_partial calls of ONE shape (instance norm: the two regression heads' norms of one depth) as one launch. */
int pwr_norm_finalize_partial_pair(const float* partial_a, const float* gamma_a, const float* beta_a, float* state_a, const float* partial_b,
                                   const float* gamma_b, const float* beta_b, float* state_b, int chunks, int B, int HW, int C, float eps,
                                   void* stream);
/* two pwr_norm_bwd_from_partial calls of ONE shape (instance norm, no addend: the two regression heads' norms of one depth, model.py:54-65 /
 * :103-114) as two launches instead of four; S1 / S2: 2 x B x C floats of scratch.  Bit-identical to the two single calls. */
int pwr_norm_bwd_from_partial_pair(const void* ga, const void* ya, const float* state_a, const float* partial_a, void* dya, float* dgamma_a,
                                   float* dbeta_a, const void* gb, const void* yb, const float* state_b, const float* partial_b, void* dyb,
                                   float* dgamma_b, float* dbeta_b, int chunks, float* S1, float* S2, int accumulate, int relu, int B, int HW,
                                   int C, int dtype, void* stream);
/* pwr_norm_bwd given the reductions of pwr_conv_fwd_stats (nb_partial). */
int pwr_norm_bwd_from_partial(const void* g, const void* y, const float* state, const float* partial, int chunks, float* S1, float* S2,
                              const void* addend, void* dy, float* dgamma, float* dbeta, int accumulate, int relu, int B, int HW,
                              int C, int mode, int dtype, void* stream);
/* Round 6, instance norm: pwr_norm_bwd_from_partial with the reduction launch taken off the caller's stream.  ONE launch: every workgroup
 * of the apply step sums the `chunks` slab rows of its sample itself (all loads in flight at once, added in pwr_norm_bwd_from_partial's
 * order -- dy is the same bits); gb != NULL: a second tensor of the same shape in the same launch (the two regression heads' norms of one
 * depth, model.py:54-65 / :103-114; no addend).  C <= 512.  The parameter gradients: pwr_norm_bwd_params_from_partial on the same slab. */
int pwr_norm_bwd_apply_from_partial(const void* ga, const void* ya, const float* state_a, const float* partial_a, const void* addend_a, void* dya,
                                    const void* gb, const void* yb, const float* state_b, const float* partial_b, void* dyb, int chunks,
                                    int relu, int B, int HW, int C, int dtype, void* stream);
/* dgamma / dbeta [C] (+)= from the slab of pwr_conv_fwd_stats (nb_partial), instance norm: the reduction launch of
 * pwr_norm_bwd_from_partial without its per-sample outputs (same order, same bits); partial_b != NULL: a second job in the same launch. */
int pwr_norm_bwd_params_from_partial(const float* partial_a, float* dgamma_a, float* dbeta_a, const float* partial_b, float* dgamma_b,
                                     float* dbeta_b, int chunks, int accumulate, int B, int HW, int C, void* stream);
/* the same for the norms of several layers in ONE launch (a backward segment's worth: the engine issues it once per segment on a side stream) */
typedef struct { const float* partial; float* dgamma; float* dbeta; int HW, C, chunks; } pwr_norm_param_job;
int pwr_norm_bwd_params_group(const pwr_norm_param_job* jobs, int njobs, int accumulate, int B, void* stream);
/* out [B,HW,C] (activation dtype) = relu(norm(y)) as the convs / weight gradients build it on operand load (fmaf(y - mean, scale, beta), ReLU,
 * one rounding): the engine materialises the heads' operands once for their weight gradients (round 6) */
int pwr_norm_apply(const void* y, const float* state, void* out, int relu, int B, int HW, int C, int dtype, void* stream);
/* dy = d/dy relu(norm(y)) applied to g (+ addend); dgamma/dbeta [C] (+)=.  S1,S2: [B,C] scratch. */
int pwr_norm_bwd(const void* g, const void* y, const float* state, float* partial, float* S1, float* S2, const void* addend, void* dy, float* dgamma,
                 float* dbeta, int accumulate, int relu, int B, int HW, int C, int mode, int dtype, void* stream);

/* Dense training targets on the device (datasets.py:285-294 heat maps = Gaussian blur, default border, of the bilinear 2x2
 * splat utils.py:37-64; datasets.py:365-383 depth-offset maps) from the normalised joints uvd [B,J,3], label_img and mask
 * [B,P,P]: heatmaps / depthmaps [B,J,P,P] fp32, the alpha < 1 targets of the loss (train.py:197-198).  ksize odd <= 15. */
int pwr_make_targets(const float* uvd, const float* label_img, const float* mask, float* heatmaps, float* depthmaps, int B, int J,
                     int P, int ksize, float sigma, void* stream);

/* ---------------------------------------------------------------------------------------------
 * ResBlock (model.py:6-23: norm, ReLU, conv1x1 C->C/2, norm, ReLU, conv3x3, norm, ReLU, conv1x1 C/2->C, + x) on the small
 * square maps of the inner hourglass levels as ONE launch per direction: one workgroup owns one sample, activations stay in
 * LDS between the three GEMMs.  bf16, InstanceNorm, C == 128, H == W in {2,4,8,16} (pwr_resblock_small_supported).
 * Forward writes what the unfused sequence (pwr_norm_stats + pwr_conv_fwd, three times) writes: the pre-norm conv outputs
 * t1, t2 [B,H,W,C/2] (NULL to skip), the three [4][B][C] norm states and out [B,H,W,C].  wa/wb/wc: kind-0 weight packs.
 * Backward takes g_out = dL/d out and writes dt2, dt1 (the dy operands of the weight gradients of conv b / conv a), dx
 * (skip connection included), the per-sample norm sums [B][2][C] and (bias_sums != NULL) the per-sample column sums of g_out
 * [B][C], all reduced over the batch by pwr_resblock_param_grads; w*_d: kind-1 packs.
 * ------------------------------------------------------------------------------------------- */
/* pwr_resblock_fwd_small with the producer of its input fused into the load: xmode 1: x = maxpool2x2(xa), xa [B,2H,2W,C] (model.py:40);
 * xmode 2: x = nearest-upsample(xh) + xa, xh [B,H/2,W/2,C], xa [B,H,W,C] (model.py:45-47), H >= 4.  x is WRITTEN (the backward pass and the
 * weight gradients read it), bit-identical to pwr_maxpool_fwd / pwr_upsample_add_fwd followed by pwr_resblock_fwd_small.  xmode 0: as
 * pwr_resblock_fwd_small. */
int pwr_resblock_fwd_small_x(int xmode, const void* xa, const void* xh, void* x, void* t1, void* t2, void* out, const void* wa,
                             const void* wb, const void* wc, const float* bias_a, const float* bias_b, const float* bias_c,
                             const float* gamma_a, const float* beta_a, const float* gamma_b, const float* beta_b,
                             const float* gamma_c, const float* beta_c, float* state_a, float* state_b, float* state_c, int B,
                             int H, int W, int C, float eps, int dtype, void* stream);
int pwr_resblock_small_supported(int H, int W, int C, int norm_mode, int dtype);
int pwr_resblock_fwd_small(const void* x, void* t1, void* t2, void* out, const void* wa, const void* wb, const void* wc,
                           const float* bias_a, const float* bias_b, const float* bias_c, const float* gamma_a,
                           const float* beta_a, const float* gamma_b, const float* beta_b, const float* gamma_c,
                           const float* beta_c, float* state_a, float* state_b, float* state_c, int B, int H, int W, int C,
                           float eps, int dtype, void* stream);
int pwr_resblock_bwd_small(const void* gout, const void* x, const void* t1, const void* t2, void* dx, void* dt1, void* dt2,
                           const void* wc_d, const void* wb_d, const void* wa_d, const float* state_a, const float* state_b,
                           const float* state_c, float* sums_a, float* sums_b, float* sums_c, float* bias_sums, int B, int H, int W,
                           int C, int dtype, void* stream);
/* pwr_resblock_bwd_small with its neighbours in the hourglass backward fused in (model.py:40-47), like pwr_resblock_fwd_small_x:
 * up_src != NULL: gout (WRITTEN) = the 2x2 block sums of up_src [B,2H,2W,C] -- pwr_upsample_bwd in the load;
 * pool_dst != NULL: pool_dst [B,2H,2W,C] = pool_addend + dx routed to the first maximum of each 2x2 window of pool_a (x = maxpool2x2(pool_a))
 * -- pwr_maxpool_bwd in the store.  Bit-identical to the separate launches. */
int pwr_resblock_bwd_small_x(const void* up_src, const void* pool_a, const void* pool_addend, void* pool_dst, void* gout, const void* x,
                             const void* t1, const void* t2, void* dx, void* dt1, void* dt2, const void* wc_d, const void* wb_d,
                             const void* wa_d, const float* state_a, const float* state_b, const float* state_c, float* sums_a,
                             float* sums_b, float* sums_c, float* bias_sums, int B, int H, int W, int C, int dtype, void* stream);
/* batch reduction of the per-sample sums of pwr_resblock_bwd_small in one launch: dgamma / dbeta of the three norms and, from
 * bias_sums [B][C] (per-sample column sums of g_out; NULL to skip), the bias gradient of conv c */
int pwr_resblock_param_grads(const float* sums_a, const float* sums_b, const float* sums_c, const float* bias_sums, float* dgamma_a,
                             float* dbeta_a, float* dgamma_b, float* dbeta_b, float* dgamma_c, float* dbeta_c, float* dbias_c,
                             int B, int C, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Hourglass plumbing (model.py:40, :45-47), NHWC.
 * ------------------------------------------------------------------------------------------- */
int pwr_maxpool_fwd(const void* x, void* y, int B, int H, int W, int C, int dtype, void* stream);
int pwr_maxpool_bwd(const void* x, const void* dh, const void* addend, void* dx, int B, int H, int W, int C, int dtype,
                    void* stream);
int pwr_upsample_add_fwd(const void* h, const void* skip, void* out, int B, int Hi, int Wi, int Ho, int Wo, int C,
                         int dtype, void* stream);
int pwr_upsample_bwd(const void* dout, void* dh, int B, int Hi, int Wi, int Ho, int Wo, int C, int dtype, void* stream);
/* fp32 [B,J,N] -> `dtype` [B,N,Jp] with channels >= J zero (feeds decoder gradients to the head convs) */
int pwr_nchw_to_nhwc_pad(const float* src, void* dst, int B, int J, int N, int Jp, int dtype, void* stream);
/* two of them (one shape) in one launch: the two regression heads' output gradients of a stage (round 6) */
int pwr_nchw_to_nhwc_pad_pair(const float* src_a, void* dst_a, const float* src_b, void* dst_b, int B, int J, int N, int Jp, int dtype, void* stream);
int pwr_add_inplace(const void* x, void* y, long long n, int dtype, void* stream);
/* bias gradients: out[c] (+)= sum_m x[m][c] for NHWC [M,C] (slab: pwr_colsum_blocks(M)*C floats); and
 * out[j] (+)= sum_{b,n} x[b][j][n] for fp32 NCHW planes */
int pwr_colsum_blocks(long long M);
int pwr_colsum_nhwc(const void* x, float* slab, float* out, long long M, int C, int accumulate, int dtype, void* stream);
int pwr_planesum_nchw(const float* x, float* part, float* out, int B, int J, int N, int accumulate, void* stream); /* part: B*J floats */
/* The concat of model.py:208, cat[heatmaps(J), depthmaps(J), label_img(1)], as an NHWC tensor [B,N,Cp] (Cp = 2J+1 rounded up
 * to a multiple of 8, extra channels zero) so that the stage-input 1x1 conv of stages >= 1 (model.py:137) runs on the matrix
 * cores like every other conv; and the inverse for its gradient (gp / gd: fp32 [B,J,N], label channel dropped). */
int pwr_cat_to_nhwc(const float* pmap, const float* dmap, const float* label, void* dst, int B, int J, int N, int Cp, int dtype,
                    void* stream);
int pwr_nhwc_to_cat_grad(const void* src, float* gp, float* gd, int B, int J, int N, int Cp, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Network engine: PixelwiseRegression.forward (model.py:200-210) and its backward as ONE static launch plan
 * per (config, batch, dtype, training).  Python makes one call per forward / backward segment.
 * ------------------------------------------------------------------------------------------- */
const char* pwr_last_error(void);

/* cfg: 8 ints {joints, stage, label_size, features, level, kernel_size, norm (0 instance, 1 batch),
 * heatmap_method}.  param_off/param_numel: offset (floats into the flat parameter buffer) and element count of
 * every parameter in the reference's named_parameters() order.  buffer_off: offsets (floats into the flat
 * buffer of BatchNorm running statistics) of running_mean, running_var per norm layer in module order
 * (NULL / 0 for instance norm).  Host pointers.  Returns NULL on error (see pwr_last_error). */
void* pwr_engine_create(const int* cfg, int B, int dtype, int training, const long long* param_off,
                        const long long* param_numel, int n_params, const long long* buffer_off, int n_buffers);
void pwr_engine_destroy(void* engine);
size_t pwr_engine_arena_bytes(void* engine);   /* activation + gradient + scratch arena the caller must provide */
size_t pwr_engine_pack_bytes(void* engine);    /* packed-weight buffer (includes the descriptor table) */
size_t pwr_engine_desc_offset(void* engine);   /* where in the pack buffer the descriptor table must be uploaded */
size_t pwr_engine_desc_bytes(void* engine);
int pwr_engine_get_descs(void* engine, void* host_dst);
int pwr_engine_num_segments(void* engine);     /* backward segments: stage S-1, ..., stage 0, stem */
int pwr_engine_num_launch_ops(void* engine, int which); /* 0 forward, 1 backward: number of fused launch groups */
/* arena, packs, flat params, flat grads, flat BN buffers (device pointers, owned by the caller) */
int pwr_engine_bind(void* engine, void* arena, void* packs, const float* params, float* grads, float* buffers);
int pwr_engine_pack(void* engine, void* stream);
/* the same re-pack as part of the NEXT pwr_engine_forward (round 6): on a side stream beside the stem's first conv -- which reads the fp32
 * parameters -- and its norm statistics, ordered behind everything already on the forward's stream; the forward waits for it in front of
 * the first launch that reads a pack */
int pwr_engine_pack_beside_forward(void* engine);
/* img [B,1,S,S], label_img / mask [B,1,P,P] fp32.  outs: HOST array of 3*stage device pointers
 * {heatmaps [B,J,P,P], depthmaps [B,J,P,P], uvd [B,J,3]} per stage (fp32), written by the call. */
int pwr_engine_forward(void* engine, const float* img, const float* label, const float* mask, void* const* outs,
                       int training, void* stream);
long long pwr_engine_generation(void* engine);
/* gouts: HOST array of 3*stage device pointers with the gradients of the outputs (NULL = zero).  Runs backward
 * segment `seg`; segment 0 first zeroes grads[0:n_grad_floats].  Parameter gradients are written (not
 * accumulated) into the bound flat gradient buffer, in the layout of the parameters. */
int pwr_engine_backward(void* engine, const void* const* gouts, int seg, long long n_grad_floats, void* stream);
/* The backward segments leave `stream` with the DATA gradients in order, but the parameter-gradient kernels run on the engine's side
 * streams; `stream` waits for them after the LAST segment only.  A consumer of an earlier segment's parameter gradients (the all-reduce
 * of that segment's slice in the data-parallel mode, ddp.py) calls this right after the segment was issued: `waiter` then waits for
 * everything issued so far on `chain_stream` and on the side streams. */
int pwr_engine_wait_segment(void* engine, void* chain_stream, void* waiter);

/* ---------------------------------------------------------------------------------------------
 * Train-step tail (train.py:139-142, 195-208) on the flat fp32 buffers.
 * ------------------------------------------------------------------------------------------- */
int pwr_loss_blocks(long long n);
/* One term of the loss of train.py:197-199: loss[0] (+)= scale * sum (a - t)^2 and, if g != NULL, g = 2*scale*(a - t).
 * scale carries alpha / lambda and the 1/(B*J) of the batch-joint mean.  partial: pwr_loss_blocks(n) floats. */
int pwr_loss_sqdiff(const float* a, const float* t, float* g, float scale, float* partial, float* loss, int accumulate, long long n,
                    void* stream);
/* torch.optim.AdamW (train.py:140 builds AdamW for --opt adam) / torch.optim.SGD(momentum) steps on flat buffers.
 * grad_scale multiplies the gradient first (1/world size after an all-reduce SUM). step is 1-based. */
int pwr_adamw_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                   float weight_decay, int step, float grad_scale, void* stream);
int pwr_sgd_step(float* p, const float* g, float* buf, long long n, float lr, float momentum, float weight_decay, int first,
                 float grad_scale, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Input pipeline on the device: /root/reference/datasets.py:243-299 for a batch of raw depth frames
 * already in HBM.  OpenCV calls restated from the published algorithms (oracle/preprocess_ref.py is the CPU twin).
 * ------------------------------------------------------------------------------------------- */
/* datasets.py:243-252 + cv2.resize (:296): depth [B,H,W] fp32 raw frames; geo [B,3] int = {first row, first column, side} of the
 * crop window (utils.center_crop, utils.py:167-173; the window may leave the frame: zeros); com_z, cube [B] double; out [B,S,S]
 * fp32 = resized crop in millimetres relative to the COM (0 = background / outside the cube). */
int pwr_crop_resize(const float* depth, const int* geo, const double* com_z, const double* cube, float* out, int B, int H, int W, int S,
                    void* stream);
/* utils.random_rotated (utils.py:66-82) image part + the `img_resize * scale` of datasets.py:282: dst = cv2.warpAffine(src, M) * scale;
 * minv [B,6] double = the INVERTED 2x3 matrices (dst -> src), scale [B] float; src, dst [B,S,S], not aliased. */
int pwr_warp_affine(const float* src, const double* minv, const float* scale, float* dst, int B, int S, void* stream);
/* datasets.py:297-299, 378-380: label = cv2.resize(img, (P,P)); mask = label != 0; img_n = img / cube; label_n = label / cube.
 * img [B,S,S]; cube [B] float; img_n [B,S,S]; label_n, mask [B,P,P]. */
int pwr_label_mask_normalize(const float* img, const float* cube, float* img_n, float* label_n, float* mask, int B, int S, int P,
                             void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PWR_H_ */
