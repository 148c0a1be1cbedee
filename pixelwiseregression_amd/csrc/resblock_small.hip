// ResBlock (/root/reference/model.py:6-23) of the inner hourglass levels as ONE kernel per direction.
//
//   out = x + conv1x1_c( relu(IN_c( conv3x3_b( relu(IN_b( conv1x1_a( relu(IN_a(x)) ) )) ) )) )      C -> C/2 -> C/2 -> C,  C = 128
//
// On the maps of 16x16 pixels and below (levels 2..5 of the hourglass, model.py:25-47) the six launches of the forward
// (3 x norm statistics + 3 x conv) and the twelve of the backward (3 x data gradient + 3 x 3 norm-backward) are pure
// latency: ~4-10 us each for microseconds of arithmetic, 16 such blocks per network.  With InstanceNorm the statistics are
// per sample, and a whole sample (<= 256 pixels x 128 channels bf16 = 64 KB) fits in the LDS of one CU, so one workgroup
// owns one sample and walks the whole block: activations never leave LDS between the three GEMMs, weights arrive by
// LDS-DMA while the previous phase's norm runs, and every global access is a coalesced 16-byte vector.
//
// The kernels write exactly what the unfused path writes (pre-norm conv outputs t1, t2, the [4][B][C] norm states, the
// data gradients dt2, dt1, dx and the per-sample [B][2][C] norm sums), with the same bf16 rounding points, so weight
// gradients (side stream) and the rest of the engine are unchanged.  bf16 + InstanceNorm only; anything else takes the
// unfused path.
#include <cstdlib>

#include "conv_common.h"
#include "pwr.h"

namespace pwr {

struct RbFwdParams {
  const bf16_t* x; bf16_t* t1; bf16_t* t2; bf16_t* out;
  const char* wa; const char* wb; const char* wc;            // kind-0 packs [tap][kch][N][64 B]
  const float* ba; const float* bb; const float* bc;
  const float* ga; const float* bta; const float* gb; const float* btb; const float* gc; const float* btc;
  float* sa; float* sb; float* sc;                            // [4][B][C] norm states (written)
  int B; float eps;
  // xmode 0: the block input is x.  1: x = maxpool2x2(xa), xa [B,2W,2W,C] (model.py:40, the hourglass level's down_sample).
  // 2: x = nearest-upsample(xh) + xa, xh [B,W/2,W/2,C], xa [B,W,W,C] (model.py:45-47).  In modes 1 / 2 the kernel computes x on the fly
  // (the arithmetic of maxpool_fwd_kernel / upsample_add_kernel) and ALSO writes it to `x` (xw), where the backward pass expects it:
  // one launch of 5 - 7 us less on the chain per level.
  int xmode = 0;
  const bf16_t* xa = nullptr; const bf16_t* xh = nullptr; bf16_t* xw = nullptr;
  long long* stamps = nullptr;      // (debug build: s_memtime of thread 0 at the phase boundaries, 16 x int64 per workgroup; pwr_debug_set_stamps)
};
struct RbBwdParams {
  const bf16_t* gout; const bf16_t* x; const bf16_t* t1; const bf16_t* t2;
  bf16_t* dx; bf16_t* dt1; bf16_t* dt2;
  const char* wcd; const char* wbd; const char* wad;         // kind-1 (data-gradient) packs
  const float* sa; const float* sb; const float* sc;
  float* sums_a; float* sums_b; float* sums_c;                // [B][2][C] per-sample (sum g, sum g*xhat)
  float* bias_sums;                                           // [B][C] per-sample column sums of g_out (bias gradient of conv c), or null
  int B;
  // The neighbours of the block in the hourglass backward (model.py:40-47), fused like their forward counterparts (round 4):
  // up_src != null: g_out = the 2x2 block sums of up_src [B,2W,2W,C] (the gradient of `nearest-upsample(h2) + a` w.r.t. h2: the arithmetic of
  //   upsample_bwd_kernel, fp32 sums in scan order, one rounding), computed while it is loaded and WRITTEN to gout_w (= gout: the weight
  //   gradient of conv c reads it; every thread reads back at the end exactly the (pixel, slot) pairs it wrote);
  // pool_dst != null: the block input is maxpool2x2(pool_a [B,2W,2W,C]); besides dx the kernel writes pool_dst = pool_addend + route(dx)
  //   (maxpool_bwd_kernel: the gradient goes to the first maximum of each window in scan order).
  const bf16_t* up_src = nullptr; bf16_t* gout_w = nullptr;
  const bf16_t* pool_a = nullptr; const bf16_t* pool_addend = nullptr; bf16_t* pool_dst = nullptr;
  long long* stamps = nullptr;      // (debug build, as in RbFwdParams)
};
#ifdef PWR_DEBUG_BUILD
long long* wstat_stamps();        // conv_patch.hip: the buffer of pwr_debug_set_stamps
#define RB_STAMP(p, i) do { if ((p).stamps && threadIdx.x == 0) (p).stamps[(size_t)blockIdx.x * 16 + (i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define RB_STAMP(p, i) do { } while (0)
#endif

constexpr int rb_max(int a, int b) { return a > b ? a : b; }

template <int LOGW>
struct RbGeom {
  static constexpr int W = 1 << LOGW, HW = W * W;
  static constexpr int MF = HW >= 32 ? HW / 32 : 1, ROWS = MF * 32;   // GEMM rows = pixels, padded to one MFMA tile
  static constexpr int P128 = 272, P64 = 144;                         // LDS row pitch: channels * 2 B + 16 B (bank spread)
  static constexpr int PW = W + 2;
  static constexpr int R_PATCH = rb_max(PW * PW * P64, ROWS * P64);   // 3x3 input patch with halo | 64-channel tile
  static constexpr int T64 = ROWS * P64;                              // raw GEMM output, 64 channels
  static constexpr int R_BYTES = rb_max(ROWS * P128, R_PATCH + T64);  // 128-channel tile overlays both
  static constexpr int W_BYTES = rb_max(18 * 64 * 64, ROWS * P128);   // all 18 tiles of the 3x3 | raw 128-channel output
  static constexpr int RED_BYTES = 4 * 128 * 2 * 4;
  // forward: gamma / beta of the three norms (rb_params_to_lds, 2 KiB); backward: the sample's norm states (rb_states_store: 4 KiB, or 2 KiB
  // without norm a's on the 16x16 maps, whose other regions leave 2496 bytes)
  static constexpr int PRM_BYTES = LOGW == 4 ? 2048 : 4096;
  static constexpr int TOTAL = R_BYTES + W_BYTES + RED_BYTES + PRM_BYTES;
  static_assert(TOTAL <= 160 * 1024, "one workgroup's LDS");
};

// The workgroup has NW waves: 8 for the 16x16 and 8x8 maps, 4 below (round 2: with 4 waves = one per SIMD nothing hid the LDS /
// global round trips of the twelve dependent phases and the element-wise passes ran on a quarter of the CU's lanes; 16 waves cap the
// kernel at 128 VGPRs and it spills ~200 of them -- measured 8: train step 6.88 -> 6.77 ms, inference 15.9k -> 16.15k frames/s).
// waves of the workgroup over an MF x NF grid of 32x32 MFMA tiles
template <int MF, int NF, int NW>
struct RbWaves {
  static constexpr int WMv = MF >= NW ? NW : MF, WNv = NW / WMv;
  static constexpr int MR = MF / WMv, NR = NF >= WNv ? NF / WNv : 1;
};

// The helpers and bodies exist twice: `rbp` reads threadIdx.x directly, `rbo` takes the work-item id as an OPAQUE value (an empty
// asm volatile), so that every helper derives its lane / wave / slot indices from its own copy and nothing derived from the id can
// be shared between helpers and kept live across the GEMM phases.
// At the 256-register limit of a 512-thread workgroup each such value was a spill: the 16x16 backward kernel went from 267 spilled
// registers to none (69.2 -> 31.8 us, same box).  The kernels that did not spill
// are ~5 % slower with the opaque id (recomputed indices, wave index through v_readfirstlane) and keep the plain one.
// -DPWR_RB_OPAQUE_TID=0 builds everything plain, for the A/B.
#ifndef PWR_RB_OPAQUE_TID
#define PWR_RB_OPAQUE_TID 1
#endif
#define RB_NS rbp
#define RB_OPAQUE 0
#include "resblock_small_body.inc"
#undef RB_NS
#undef RB_OPAQUE
#define RB_NS rbo
#define RB_OPAQUE PWR_RB_OPAQUE_TID
#include "resblock_small_body.inc"
#undef RB_NS
#undef RB_OPAQUE

template <int LOGW, int NW>
__global__ __launch_bounds__(64 * NW) void resblock_fwd_small_kernel(RbFwdParams p) {
  __shared__ __attribute__((aligned(16))) char smem[RbGeom<LOGW>::TOTAL];
  rbp::rb_fwd_body<LOGW, NW>(p, blockIdx.x, smem);
}

template <int LOGW, int NW>
__global__ __launch_bounds__(64 * NW) void resblock_bwd_small_kernel(RbBwdParams p) {
  __shared__ __attribute__((aligned(16))) char smem[RbGeom<LOGW>::TOTAL];
  if constexpr (LOGW == 4 && NW == 8) rbo::rb_bwd_body<LOGW, NW>(p, blockIdx.x, smem);   // (the one that spilled)
  else rbp::rb_bwd_body<LOGW, NW>(p, blockIdx.x, smem);
}

// dst[c] = sum_b src[b * stride + c]  (fixed order), one block per job: the three norms' dgamma / dbeta and conv c's bias
struct RbPgJobs { const float* src[7]; float* dst[7]; int stride[7]; int C[7]; int B; };
__global__ __launch_bounds__(128) void rb_param_grad_kernel(RbPgJobs j) {
  const int k = blockIdx.x, c = threadIdx.x;
  if (!j.dst[k] || c >= j.C[k]) return;
  const float* s = j.src[k] + c;
  const int st = j.stride[k];
  float t = 0.f;
  int b = 0;
  for (; b + 8 <= j.B; b += 8) {
    float a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = s[(size_t)(b + u) * st];
#pragma unroll
    for (int u = 0; u < 8; ++u) t += a[u];
  }
  for (; b < j.B; ++b) t += s[(size_t)b * st];
  j.dst[k][c] = t;
}

}  // namespace pwr

using namespace pwr;

// Batch reduction of everything pwr_resblock_bwd_small left per sample: dgamma / dbeta of the three norms ([B][2][C] sums) and
// the bias gradient of conv c (bias_sums [B][C], NULL to skip).  One launch.
extern "C" int pwr_resblock_param_grads(const float* sums_a, const float* sums_b, const float* sums_c, const float* bias_sums,
                                        float* dgamma_a, float* dbeta_a, float* dgamma_b, float* dbeta_b, float* dgamma_c,
                                        float* dbeta_c, float* dbias_c, int B, int C, void* stream) {
  if (C > 128) return PWR_EUNSUPPORTED;
  const int Fh = C / 2;
  RbPgJobs j;
  const float* srcs[7] = {sums_a + C, sums_a, sums_b + Fh, sums_b, sums_c + Fh, sums_c, bias_sums};
  float* dsts[7] = {dgamma_a, dbeta_a, dgamma_b, dbeta_b, dgamma_c, dbeta_c, bias_sums ? dbias_c : nullptr};
  const int strides[7] = {2 * C, 2 * C, 2 * Fh, 2 * Fh, 2 * Fh, 2 * Fh, C};
  const int Cs[7] = {C, C, Fh, Fh, Fh, Fh, C};
  for (int k = 0; k < 7; ++k) { j.src[k] = srcs[k]; j.dst[k] = dsts[k]; j.stride[k] = strides[k]; j.C[k] = Cs[k]; }
  j.B = B;
  hipLaunchKernelGGL(rb_param_grad_kernel, dim3(7), dim3(128), 0, (hipStream_t)stream, j);
  return (int)hipGetLastError();
}

extern "C" int pwr_resblock_small_supported(int H, int W, int C, int norm_mode, int dtype) {
  static const bool on = (PWR_DBG_ENV("PWR_RESBLOCK_FUSED", 1) != 0);
  return on && dtype == PWR_BF16 && norm_mode == 0 && C == 128 && H == W && (W == 2 || W == 4 || W == 8 || W == 16);
}

static int rb_fwd_launch(const RbFwdParams& p, int W, hipStream_t s) {
  const int B = p.B;
  const int wide = PWR_DBG_ENV("PWR_RESBLOCK_WAVES", 1);
  // 0: four waves everywhere (round 1); 1: eight on the 16x16 / 8x8 maps (measured best); 2: eight everywhere.
  if (W == 16 && wide) hipLaunchKernelGGL((resblock_fwd_small_kernel<4, 8>), dim3(B), dim3(512), 0, s, p);
  else if (W == 16) hipLaunchKernelGGL((resblock_fwd_small_kernel<4, 4>), dim3(B), dim3(256), 0, s, p);
  else if (W == 8 && wide) hipLaunchKernelGGL((resblock_fwd_small_kernel<3, 8>), dim3(B), dim3(512), 0, s, p);
  else if (W == 8) hipLaunchKernelGGL((resblock_fwd_small_kernel<3, 4>), dim3(B), dim3(256), 0, s, p);
  else if (W == 4 && wide == 2) hipLaunchKernelGGL((resblock_fwd_small_kernel<2, 8>), dim3(B), dim3(512), 0, s, p);
  else if (W == 4) hipLaunchKernelGGL((resblock_fwd_small_kernel<2, 4>), dim3(B), dim3(256), 0, s, p);
  else if (wide == 2) hipLaunchKernelGGL((resblock_fwd_small_kernel<1, 8>), dim3(B), dim3(512), 0, s, p);
  else hipLaunchKernelGGL((resblock_fwd_small_kernel<1, 4>), dim3(B), dim3(256), 0, s, p);
  return (int)hipGetLastError();
}

// xmode 0: x is the block input.  1: x (written) = maxpool2x2(xa [B,2H,2W,C]).  2: x (written) = nearest-upsample(xh [B,H/2,W/2,C]) + xa [B,H,W,C].
extern "C" int pwr_resblock_fwd_small_x(int xmode, const void* xa, const void* xh, void* x, void* t1, void* t2, void* out, const void* wa,
                                        const void* wb, const void* wc, const float* bias_a, const float* bias_b, const float* bias_c,
                                        const float* gamma_a, const float* beta_a, const float* gamma_b, const float* beta_b,
                                        const float* gamma_c, const float* beta_c, float* state_a, float* state_b, float* state_c, int B,
                                        int H, int W, int C, float eps, int dtype, void* stream) {
  if (!(dtype == PWR_BF16 && C == 128 && H == W && (W == 2 || W == 4 || W == 8 || W == 16))) return (int)hipErrorInvalidValue;
  if (xmode < 0 || xmode > 2 || (xmode == 1 && !xa) || (xmode == 2 && (!xa || !xh || W < 4)) || !x) return PWR_EINVAL;
  RbFwdParams p;
  p.x = (const bf16_t*)x; p.t1 = (bf16_t*)t1; p.t2 = (bf16_t*)t2; p.out = (bf16_t*)out;
  p.wa = (const char*)wa; p.wb = (const char*)wb; p.wc = (const char*)wc;
  p.ba = bias_a; p.bb = bias_b; p.bc = bias_c;
  p.ga = gamma_a; p.bta = beta_a; p.gb = gamma_b; p.btb = beta_b; p.gc = gamma_c; p.btc = beta_c;
  p.sa = state_a; p.sb = state_b; p.sc = state_c;
  p.B = B; p.eps = eps;
  p.xmode = xmode; p.xa = (const bf16_t*)xa; p.xh = (const bf16_t*)xh; p.xw = (bf16_t*)x;
#ifdef PWR_DEBUG_BUILD
  p.stamps = wstat_stamps();
#endif
  return rb_fwd_launch(p, W, (hipStream_t)stream);
}

extern "C" int pwr_resblock_fwd_small(const void* x, void* t1, void* t2, void* out, const void* wa, const void* wb, const void* wc,
                                      const float* bias_a, const float* bias_b, const float* bias_c, const float* gamma_a,
                                      const float* beta_a, const float* gamma_b, const float* beta_b, const float* gamma_c,
                                      const float* beta_c, float* state_a, float* state_b, float* state_c, int B, int H, int W, int C,
                                      float eps, int dtype, void* stream) {
  return pwr_resblock_fwd_small_x(0, nullptr, nullptr, const_cast<void*>(x), t1, t2, out, wa, wb, wc, bias_a, bias_b, bias_c, gamma_a, beta_a,
                                  gamma_b, beta_b, gamma_c, beta_c, state_a, state_b, state_c, B, H, W, C, eps, dtype, stream);
}

extern "C" int pwr_resblock_bwd_small(const void* gout, const void* x, const void* t1, const void* t2, void* dx, void* dt1, void* dt2,
                                      const void* wc_d, const void* wb_d, const void* wa_d, const float* state_a, const float* state_b,
                                      const float* state_c, float* sums_a, float* sums_b, float* sums_c, float* bias_sums, int B, int H,
                                      int W, int C, int dtype, void* stream) {
  return pwr_resblock_bwd_small_x(nullptr, nullptr, nullptr, nullptr, const_cast<void*>(gout), x, t1, t2, dx, dt1, dt2, wc_d, wb_d, wa_d, state_a,
                                  state_b, state_c, sums_a, sums_b, sums_c, bias_sums, B, H, W, C, dtype, stream);
}

// up_src != NULL: gout (WRITTEN) = 2x2 block sums of up_src [B,2H,2W,C] (pwr_upsample_bwd fused into the load).  pool_dst != NULL:
// pool_dst [B,2H,2W,C] = pool_addend + the block's dx routed to the first maximum of every 2x2 window of pool_a (pwr_maxpool_bwd fused
// into the store; x = maxpool2x2(pool_a)).  Everything written is bit-identical to the separate launches.
extern "C" int pwr_resblock_bwd_small_x(const void* up_src, const void* pool_a, const void* pool_addend, void* pool_dst, void* gout, const void* x,
                                        const void* t1, const void* t2, void* dx, void* dt1, void* dt2, const void* wc_d, const void* wb_d,
                                        const void* wa_d, const float* state_a, const float* state_b, const float* state_c, float* sums_a,
                                        float* sums_b, float* sums_c, float* bias_sums, int B, int H, int W, int C, int dtype, void* stream) {
  if (!(dtype == PWR_BF16 && C == 128 && H == W && (W == 2 || W == 4 || W == 8 || W == 16))) return (int)hipErrorInvalidValue;
  if (pool_dst && (!pool_a || !pool_addend)) return PWR_EINVAL;
  RbBwdParams p;
  p.up_src = (const bf16_t*)up_src; p.gout_w = (bf16_t*)gout;
  p.pool_a = (const bf16_t*)pool_a; p.pool_addend = (const bf16_t*)pool_addend; p.pool_dst = (bf16_t*)pool_dst;
  p.gout = (const bf16_t*)gout; p.x = (const bf16_t*)x; p.t1 = (const bf16_t*)t1; p.t2 = (const bf16_t*)t2;
  p.dx = (bf16_t*)dx; p.dt1 = (bf16_t*)dt1; p.dt2 = (bf16_t*)dt2;
  p.wcd = (const char*)wc_d; p.wbd = (const char*)wb_d; p.wad = (const char*)wa_d;
  p.sa = state_a; p.sb = state_b; p.sc = state_c;
  p.sums_a = sums_a; p.sums_b = sums_b; p.sums_c = sums_c; p.bias_sums = bias_sums;
  p.B = B;
#ifdef PWR_DEBUG_BUILD
  p.stamps = wstat_stamps();
#endif
  hipStream_t s = (hipStream_t)stream;
  const int wide = PWR_DBG_ENV("PWR_RESBLOCK_WAVES", 1);
  // 0: four waves everywhere (round 1); 1: eight on the 16x16 / 8x8 maps (measured best); 2: eight everywhere.
  if (W == 16 && wide) hipLaunchKernelGGL((resblock_bwd_small_kernel<4, 8>), dim3(B), dim3(512), 0, s, p);
  else if (W == 16) hipLaunchKernelGGL((resblock_bwd_small_kernel<4, 4>), dim3(B), dim3(256), 0, s, p);
  else if (W == 8 && wide) hipLaunchKernelGGL((resblock_bwd_small_kernel<3, 8>), dim3(B), dim3(512), 0, s, p);
  else if (W == 8) hipLaunchKernelGGL((resblock_bwd_small_kernel<3, 4>), dim3(B), dim3(256), 0, s, p);
  else if (W == 4 && wide == 2) hipLaunchKernelGGL((resblock_bwd_small_kernel<2, 8>), dim3(B), dim3(512), 0, s, p);
  else if (W == 4) hipLaunchKernelGGL((resblock_bwd_small_kernel<2, 4>), dim3(B), dim3(256), 0, s, p);
  else if (wide == 2) hipLaunchKernelGGL((resblock_bwd_small_kernel<1, 8>), dim3(B), dim3(512), 0, s, p);
  else hipLaunchKernelGGL((resblock_bwd_small_kernel<1, 4>), dim3(B), dim3(256), 0, s, p);
  return (int)hipGetLastError();
}

