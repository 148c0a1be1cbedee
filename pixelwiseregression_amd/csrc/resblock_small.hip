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
};
struct RbBwdParams {
  const bf16_t* gout; const bf16_t* x; const bf16_t* t1; const bf16_t* t2;
  bf16_t* dx; bf16_t* dt1; bf16_t* dt2;
  const char* wcd; const char* wbd; const char* wad;         // kind-1 (data-gradient) packs
  const float* sa; const float* sb; const float* sc;
  float* sums_a; float* sums_b; float* sums_c;                // [B][2][C] per-sample (sum g, sum g*xhat)
  float* bias_sums;                                           // [B][C] per-sample column sums of g_out (bias gradient of conv c), or null
  int B;
};

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
  static constexpr int TOTAL = R_BYTES + W_BYTES + RED_BYTES;
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

// whole weight pack -> LDS by LDS-DMA: `ntiles` tiles of [NROWS][64 B]; 1 KiB (16 rows) per wave instruction, XOR swizzle
// applied to the source address (cdna_hip_programming.md rule 21).  Completion: s_waitcnt vmcnt(0) + barrier.
template <int NROWS, int NW>
__device__ __forceinline__ void rb_dma_weights(const char* __restrict__ pack, int ntiles, char* Wl) {
  constexpr int CPT = NROWS / 16;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nch = ntiles * CPT;
  for (int c = wid; c < nch; c += NW) {
    const int tile = c / CPT, row = 16 * (c % CPT) + (lane >> 2);
    const int slot = (lane & 3) ^ ((row >> 2) & 3);
    const char* src = pack + ((size_t)tile * NROWS + row) * 64 + slot * 16;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(Wl + c * 1024), 16, 0, 0);
  }
}

__device__ __forceinline__ void rb_wait_sync() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// acc[MR][NR] = A (LDS: [pixel][K] tile of pitch AP, or the 3x3 patch) x weights (LDS tiles)
template <int LOGW, int MF, int NF, int TAPS, int KCH, bool PATCH, int AP, int NW>
__device__ __forceinline__ void rb_gemm(const char* A, const char* Wl, f32x16 (&acc)[RbWaves<MF, NF, NW>::MR][RbWaves<MF, NF, NW>::NR]) {
  typedef RbWaves<MF, NF, NW> WT;
  constexpr int MR = WT::MR, NR = WT::NR, Wd = 1 << LOGW, PWp = Wd + 2;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int wm = wid / WT::WNv, wn = wid % WT::WNv;
#pragma unroll
  for (int i = 0; i < MR; ++i)
#pragma unroll
    for (int j = 0; j < NR; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  if (wn * NR >= NF) return;
  const char* aBase[MR];
#pragma unroll
  for (int i = 0; i < MR; ++i) {
    const int R = (wm * MR + i) * 32 + r;
    aBase[i] = PATCH ? A + ((R >> LOGW) * PWp + (R & (Wd - 1))) * 144 + h * 16 : A + R * AP + h * 16;
  }
  int bOff[NR][2];
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    bOff[j][0] = lds_off((wn * NR + j) * 32 + r, h);
    bOff[j][1] = lds_off((wn * NR + j) * 32 + r, 2 + h);
  }
#pragma unroll
  for (int it = 0; it < TAPS * KCH; ++it) {
    const int tap = it / KCH, kch = it - tap * KCH;
    const int ky = tap / 3, kx = tap - ky * 3;
    const int toff = PATCH ? (ky * PWp + kx) * 144 : 0;
    const char* Wt = Wl + it * (NF * 32 * 64);
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
      bf16x8 a[MR], bb[NR];
#pragma unroll
      for (int i = 0; i < MR; ++i) a[i] = *reinterpret_cast<const bf16x8*>(aBase[i] + toff + kch * 64 + ss * 32);
#pragma unroll
      for (int j = 0; j < NR; ++j) bb[j] = *reinterpret_cast<const bf16x8*>(Wt + bOff[j][ss]);
#pragma unroll
      for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], bb[j], acc[i][j], 0, 0, 0);
    }
  }
}

// accumulators (+ bias) -> bf16 -> LDS [pixel][channel] of pitch TP (the rounding point of the unfused conv epilogue)
template <int MF, int NF, int NW>
__device__ __forceinline__ void rb_acc_to_lds(const f32x16 (&acc)[RbWaves<MF, NF, NW>::MR][RbWaves<MF, NF, NW>::NR], const float* __restrict__ bias,
                                              char* T, int TP) {
  typedef RbWaves<MF, NF, NW> WT;
  constexpr int MR = WT::MR, NR = WT::NR;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int wm = wid / WT::WNv, wn = wid % WT::WNv;
  if (wn * NR >= NF) return;
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    const int ch = (wn * NR + j) * 32 + r;
    const float bj = bias ? bias[ch] : 0.f;
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (wm * MR + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        *reinterpret_cast<bf16_t*>(T + row * TP + ch * 2) = (bf16_t)(acc[i][j][e] + bj);
      }
  }
}

// sum of s[0..15] over the pixel lanes of the workgroup (threads with the same channel slot); result in all threads.
// `red` holds 1024 floats: with more than four waves the 16 values go through it in rounds of 1024 / (NW * NSLOT).
template <int NSLOT, int NW>
__device__ __forceinline__ void rb_reduce16(float (&s)[16], float* red, int slot) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int o = NSLOT; o < 64; o <<= 1)
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] += __shfl_xor(s[e], o, 64);
  constexpr int CAP = 1024 / (NW * NSLOT), VPR = CAP >= 16 ? 16 : CAP;
  static_assert(VPR >= 1 && 16 % VPR == 0, "rounds of whole values");
#pragma unroll
  for (int r0 = 0; r0 < 16; r0 += VPR) {
    __syncthreads();   // `red` may still be read from a previous use / round
    if (lane < NSLOT) {
#pragma unroll
      for (int e = 0; e < VPR; ++e) red[(wid * NSLOT + slot) * VPR + e] = s[r0 + e];
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < VPR; ++e) {
      if constexpr (NW == 4) {
        s[r0 + e] = (red[(0 * NSLOT + slot) * VPR + e] + red[(1 * NSLOT + slot) * VPR + e]) +
                    (red[(2 * NSLOT + slot) * VPR + e] + red[(3 * NSLOT + slot) * VPR + e]);
      } else {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += red[(w * NSLOT + slot) * VPR + e];
        s[r0 + e] = t;
      }
    }
  }
}

template <int LOGW, int DST>
__device__ __forceinline__ int rb_dst_off(int px, int slot, int DP) {
  constexpr int Wd = 1 << LOGW;
  if (DST == 1) return (((px >> LOGW) + 1) * (Wd + 2) + (px & (Wd - 1)) + 1) * 144 + slot * 16;   // patch interior
  return px * DP + slot * 16;
}

// zero the halo of the 3x3 patch (W+2)^2 x 64 channels
template <int LOGW, int NW>
__device__ __forceinline__ void rb_zero_halo(char* R) {
  constexpr int PWp = (1 << LOGW) + 2;
  for (int idx = threadIdx.x; idx < PWp * PWp * 8; idx += 64 * NW) {
    const int pix = idx >> 3, s = idx & 7;
    const int py = pix / PWp, px = pix - py * PWp;
    if (py == 0 || py == PWp - 1 || px == 0 || px == PWp - 1) *reinterpret_cast<bf16x8*>(R + pix * 144 + s * 16) = bf16x8{};
  }
}

// InstanceNorm statistics of one sample's [HW][CH] map + normalise + ReLU into the next GEMM's A operand.
// Source: global (the block input) or LDS (raw conv output, also stored to global for the backward pass).
template <int CH, int LOGW, bool SRC_GLOBAL, int DST, int NW>
__device__ __forceinline__ void rb_norm_fwd(const bf16_t* __restrict__ gsrc, const char* T, int TP, bf16_t* __restrict__ raw_dst,
                                            const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ state,
                                            int b, int B, char* dst, int DP, float eps, float* red) {
  constexpr int HW = 1 << (2 * LOGW), NSLOT = CH / 8, PL = 64 * NW / NSLOT, NPX = (HW + PL - 1) / PL;
  const int slot = threadIdx.x % NSLOT, pl = threadIdx.x / NSLOT;
  float ga[8], be[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { ga[e] = gamma[slot * 8 + e]; be[e] = beta[slot * 8 + e]; }
  bf16x8 v[NPX];
  const bf16x8 k0 = SRC_GLOBAL ? *reinterpret_cast<const bf16x8*>(gsrc + slot * 8) : *reinterpret_cast<const bf16x8*>(T + slot * 16);
#pragma unroll
  for (int k = 0; k < NPX; ++k) {
    const int px = pl + k * PL;
    v[k] = bf16x8{};
    if (px < HW) {
      if (SRC_GLOBAL) v[k] = *reinterpret_cast<const bf16x8*>(gsrc + (size_t)px * CH + slot * 8);
      else v[k] = *reinterpret_cast<const bf16x8*>(T + px * TP + slot * 16);
    }
  }
  if (!SRC_GLOBAL && raw_dst) {
#pragma unroll
    for (int k = 0; k < NPX; ++k) {
      const int px = pl + k * PL;
      if (px < HW) *reinterpret_cast<bf16x8*>(raw_dst + (size_t)px * CH + slot * 8) = v[k];
    }
  }
  // shifted single pass: sums of (v - v[pixel 0]) and its square
  float s[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
  for (int k = 0; k < NPX; ++k) {
    const int px = pl + k * PL;
    if (px < HW) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = (float)v[k][e] - (float)k0[e];
        s[e] += d;
        s[8 + e] = fmaf(d, d, s[8 + e]);
      }
    }
  }
  rb_reduce16<NSLOT, NW>(s, red, slot);
  float mean[8], scale[8];
  const float inv = 1.f / (float)HW;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float m = s[e] * inv;
    float var = s[8 + e] * inv - m * m;
    var = var > 0.f ? var : 0.f;
    const float rstd = 1.f / sqrtf(var + eps);
    mean[e] = (float)k0[e] + m;
    scale[e] = rstd * ga[e];
    if (state && pl == 0) {
      const size_t plane = (size_t)B * CH, c = (size_t)b * CH + slot * 8 + e;
      state[c] = mean[e]; state[plane + c] = rstd; state[2 * plane + c] = scale[e]; state[3 * plane + c] = be[e];
    }
  }
#pragma unroll
  for (int k = 0; k < NPX; ++k) {
    const int px = pl + k * PL;
    if (px < HW) {
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16_t)fmaxf(fmaf((float)v[k][e] - mean[e], scale[e], be[e]), 0.f);
      *reinterpret_cast<bf16x8*>(dst + rb_dst_off<LOGW, DST>(px, slot, DP)) = o;
    }
  }
}

// backward of relu(IN(y)) for one sample: g (raw data gradient, LDS) -> dy = scale * (gm - mean(gm) - xhat * mean(gm * xhat))
// (+ addend), written to global (for the weight gradient) and, DST 0/1, into the next GEMM's A operand.
template <int CH, int LOGW, int DST, int NW>
__device__ __forceinline__ void rb_norm_bwd(const char* T, int TP, const bf16_t* __restrict__ ysrc, const float* __restrict__ state, int b, int B,
                                            const bf16_t* __restrict__ addend, bf16_t* __restrict__ dy_dst, float* __restrict__ sums,
                                            char* dst, int DP, float* red) {
  constexpr int HW = 1 << (2 * LOGW), NSLOT = CH / 8, PL = 64 * NW / NSLOT, NPX = (HW + PL - 1) / PL;
  const int slot = threadIdx.x % NSLOT, pl = threadIdx.x / NSLOT;
  float mu[8], rs[8], sc[8], sh[8];
  {
    const size_t plane = (size_t)B * CH, c = (size_t)b * CH + slot * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) { mu[e] = state[c + e]; rs[e] = state[plane + c + e]; sc[e] = state[2 * plane + c + e]; sh[e] = state[3 * plane + c + e]; }
  }
  bf16x8 g[NPX], y[NPX];
#pragma unroll
  for (int k = 0; k < NPX; ++k) {
    const int px = pl + k * PL;
    g[k] = bf16x8{}; y[k] = bf16x8{};
    if (px < HW) {
      y[k] = *reinterpret_cast<const bf16x8*>(ysrc + (size_t)px * CH + slot * 8);
      g[k] = *reinterpret_cast<const bf16x8*>(T + px * TP + slot * 16);
    }
  }
  float s[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
  for (int k = 0; k < NPX; ++k) {
    const int px = pl + k * PL;
    if (px < HW) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float yy = (float)y[k][e];
        float gg = (float)g[k][e];
        if (!(fmaf(yy - mu[e], sc[e], sh[e]) > 0.f)) gg = 0.f;
        g[k][e] = (bf16_t)gg;     // masked (exact: gg is a bf16 value or 0)
        s[e] += gg;
        s[8 + e] = fmaf(gg, (yy - mu[e]) * rs[e], s[8 + e]);
      }
    }
  }
  rb_reduce16<NSLOT, NW>(s, red, slot);
  if (pl == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { sums[slot * 8 + e] = s[e]; sums[CH + slot * 8 + e] = s[8 + e]; }
  }
  const float inv = 1.f / (float)HW;
#pragma unroll
  for (int e = 0; e < 16; ++e) s[e] *= inv;
#pragma unroll
  for (int k = 0; k < NPX; ++k) {
    const int px = pl + k * PL;
    if (px < HW) {
      bf16x8 av = {};
      if (addend) av = *reinterpret_cast<const bf16x8*>(addend + (size_t)px * CH + slot * 8);
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xn = ((float)y[k][e] - mu[e]) * rs[e];
        float rr = sc[e] * ((float)g[k][e] - s[e] - xn * s[8 + e]);
        if (addend) rr += (float)av[e];
        o[e] = (bf16_t)rr;
      }
      *reinterpret_cast<bf16x8*>(dy_dst + (size_t)px * CH + slot * 8) = o;
      if (DST != 2) *reinterpret_cast<bf16x8*>(dst + rb_dst_off<LOGW, DST>(px, slot, DP)) = o;
    }
  }
}

template <int LOGW, int NW>
__global__ __launch_bounds__(64 * NW) void resblock_fwd_small_kernel(RbFwdParams p) {
  typedef RbGeom<LOGW> G;
  constexpr int HW = G::HW, MF = G::MF;
  __shared__ __attribute__((aligned(16))) char smem[G::TOTAL];
  char* R = smem;
  char* T = smem + G::R_PATCH;
  char* Wl = smem + G::R_BYTES;
  float* red = reinterpret_cast<float*>(smem + G::R_BYTES + G::W_BYTES);
  const int b = blockIdx.x;

  // ---- a0 = relu(IN_a(x)) -> R [HW][128]
  rb_dma_weights<64, NW>(p.wa, 4, Wl);
  rb_norm_fwd<128, LOGW, true, 0, NW>(p.x + (size_t)b * HW * 128, nullptr, 0, nullptr, p.ga, p.bta, p.sa, b, p.B, R, G::P128, p.eps, red);
  rb_wait_sync();
  // ---- t1 = conv1x1_a(a0) + bias
  {
    f32x16 acc[RbWaves<MF, 2, NW>::MR][RbWaves<MF, 2, NW>::NR];
    rb_gemm<LOGW, MF, 2, 1, 4, false, G::P128, NW>(R, Wl, acc);
    __syncthreads();                       // R (a0) and the weights are dead
    rb_dma_weights<64, NW>(p.wb, 18, Wl);      // lands while the norm below runs
    rb_acc_to_lds<MF, 2, NW>(acc, p.ba, T, G::P64);
  }
  rb_zero_halo<LOGW, NW>(R);
  __syncthreads();
  rb_norm_fwd<64, LOGW, false, 1, NW>(nullptr, T, G::P64, p.t1 ? p.t1 + (size_t)b * HW * 64 : nullptr, p.gb, p.btb, p.sb, b, p.B, R, G::P64, p.eps, red);
  rb_wait_sync();
  // ---- t2 = conv3x3_b(a1) + bias
  {
    f32x16 acc[RbWaves<MF, 2, NW>::MR][RbWaves<MF, 2, NW>::NR];
    rb_gemm<LOGW, MF, 2, 9, 2, true, G::P64, NW>(R, Wl, acc);
    __syncthreads();
    rb_dma_weights<128, NW>(p.wc, 2, Wl);
    rb_acc_to_lds<MF, 2, NW>(acc, p.bb, T, G::P64);
  }
  __syncthreads();
  rb_norm_fwd<64, LOGW, false, 0, NW>(nullptr, T, G::P64, p.t2 ? p.t2 + (size_t)b * HW * 64 : nullptr, p.gc, p.btc, p.sc, b, p.B, R, G::P64, p.eps, red);
  rb_wait_sync();
  // ---- out = conv1x1_c(a2) + bias + x
  {
    f32x16 acc[RbWaves<MF, 4, NW>::MR][RbWaves<MF, 4, NW>::NR];
    rb_gemm<LOGW, MF, 4, 1, 2, false, G::P64, NW>(R, Wl, acc);
    __syncthreads();
    rb_acc_to_lds<MF, 4, NW>(acc, p.bc, Wl, G::P128);
  }
  __syncthreads();
  {
    const int slot = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const bf16_t* xs = p.x + (size_t)b * HW * 128;
    bf16_t* os = p.out + (size_t)b * HW * 128;
    for (int px = pl; px < HW; px += 4 * NW) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(Wl + px * G::P128 + slot * 16);
      const bf16x8 xr = *reinterpret_cast<const bf16x8*>(xs + (size_t)px * 128 + slot * 8);
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16_t)((float)v[e] + (float)xr[e]);
      *reinterpret_cast<bf16x8*>(os + (size_t)px * 128 + slot * 8) = o;
    }
  }
}

template <int LOGW, int NW>
__global__ __launch_bounds__(64 * NW) void resblock_bwd_small_kernel(RbBwdParams p) {
  typedef RbGeom<LOGW> G;
  constexpr int HW = G::HW, MF = G::MF;
  __shared__ __attribute__((aligned(16))) char smem[G::TOTAL];
  char* R = smem;
  char* T = smem + G::R_PATCH;
  char* Wl = smem + G::R_BYTES;
  float* red = reinterpret_cast<float*>(smem + G::R_BYTES + G::W_BYTES);
  const int b = blockIdx.x;
  const bf16_t* go = p.gout + (size_t)b * HW * 128;

  // ---- g_out -> R [HW][128]
  rb_dma_weights<64, NW>(p.wcd, 4, Wl);
  {
    const int slot = threadIdx.x & 15, pl = threadIdx.x >> 4;
    constexpr int PLG = 4 * NW, NPX = (HW + PLG - 1) / PLG;
    bf16x8 v[NPX];
#pragma unroll
    for (int k = 0; k < NPX; ++k) {
      const int px = pl + PLG * k;
      v[k] = bf16x8{};
      if (px < HW) v[k] = *reinterpret_cast<const bf16x8*>(go + (size_t)px * 128 + slot * 8);
    }
#pragma unroll
    for (int k = 0; k < NPX; ++k) {
      const int px = pl + PLG * k;
      if (px < HW) *reinterpret_cast<bf16x8*>(R + px * G::P128 + slot * 16) = v[k];
    }
    if (p.bias_sums) {   // db_c[b][c] = sum over pixels of g_out (conv c's bias gradient, summed over b by rb_param_grad_kernel)
      float s[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
      for (int k = 0; k < NPX; ++k) {
        const int px = pl + PLG * k;
        if (px < HW) {
#pragma unroll
          for (int e = 0; e < 8; ++e) s[e] += (float)v[k][e];
        }
      }
      rb_reduce16<16, NW>(s, red, slot);
      if (pl == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) p.bias_sums[(size_t)b * 128 + slot * 8 + e] = s[e];
      }
    }
  }
  rb_wait_sync();
  // ---- g wrt a2 = g_out x Wc^T ; norm-backward c -> dt2
  {
    f32x16 acc[RbWaves<MF, 2, NW>::MR][RbWaves<MF, 2, NW>::NR];
    rb_gemm<LOGW, MF, 2, 1, 4, false, G::P128, NW>(R, Wl, acc);
    __syncthreads();
    rb_dma_weights<64, NW>(p.wbd, 18, Wl);
    rb_acc_to_lds<MF, 2, NW>(acc, nullptr, T, G::P64);
  }
  rb_zero_halo<LOGW, NW>(R);
  __syncthreads();
  rb_norm_bwd<64, LOGW, 1, NW>(T, G::P64, p.t2 + (size_t)b * HW * 64, p.sc, b, p.B, nullptr, p.dt2 + (size_t)b * HW * 64,
                           p.sums_c + (size_t)b * 2 * 64, R, G::P64, red);
  rb_wait_sync();
  // ---- g wrt a1 = conv3x3(dt2, flipped Wb) ; norm-backward b -> dt1
  {
    f32x16 acc[RbWaves<MF, 2, NW>::MR][RbWaves<MF, 2, NW>::NR];
    rb_gemm<LOGW, MF, 2, 9, 2, true, G::P64, NW>(R, Wl, acc);
    __syncthreads();
    rb_dma_weights<128, NW>(p.wad, 2, Wl);
    rb_acc_to_lds<MF, 2, NW>(acc, nullptr, T, G::P64);
  }
  __syncthreads();
  rb_norm_bwd<64, LOGW, 0, NW>(T, G::P64, p.t1 + (size_t)b * HW * 64, p.sb, b, p.B, nullptr, p.dt1 + (size_t)b * HW * 64,
                           p.sums_b + (size_t)b * 2 * 64, R, G::P64, red);
  rb_wait_sync();
  // ---- g wrt a0 = dt1 x Wa^T ; norm-backward a + skip -> dx
  {
    f32x16 acc[RbWaves<MF, 4, NW>::MR][RbWaves<MF, 4, NW>::NR];
    rb_gemm<LOGW, MF, 4, 1, 2, false, G::P64, NW>(R, Wl, acc);
    __syncthreads();
    rb_acc_to_lds<MF, 4, NW>(acc, nullptr, Wl, G::P128);
  }
  __syncthreads();
  rb_norm_bwd<128, LOGW, 2, NW>(Wl, G::P128, p.x + (size_t)b * HW * 128, p.sa, b, p.B, go, p.dx + (size_t)b * HW * 128,
                            p.sums_a + (size_t)b * 2 * 128, nullptr, 0, red);
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
  static const bool on = [] { const char* e = getenv("PWR_RESBLOCK_FUSED"); return e ? atoi(e) != 0 : true; }();
  return on && dtype == PWR_BF16 && norm_mode == 0 && C == 128 && H == W && (W == 2 || W == 4 || W == 8 || W == 16);
}

extern "C" int pwr_resblock_fwd_small(const void* x, void* t1, void* t2, void* out, const void* wa, const void* wb, const void* wc,
                                      const float* bias_a, const float* bias_b, const float* bias_c, const float* gamma_a,
                                      const float* beta_a, const float* gamma_b, const float* beta_b, const float* gamma_c,
                                      const float* beta_c, float* state_a, float* state_b, float* state_c, int B, int H, int W, int C,
                                      float eps, int dtype, void* stream) {
  if (!(dtype == PWR_BF16 && C == 128 && H == W && (W == 2 || W == 4 || W == 8 || W == 16))) return (int)hipErrorInvalidValue;
  RbFwdParams p;
  p.x = (const bf16_t*)x; p.t1 = (bf16_t*)t1; p.t2 = (bf16_t*)t2; p.out = (bf16_t*)out;
  p.wa = (const char*)wa; p.wb = (const char*)wb; p.wc = (const char*)wc;
  p.ba = bias_a; p.bb = bias_b; p.bc = bias_c;
  p.ga = gamma_a; p.bta = beta_a; p.gb = gamma_b; p.btb = beta_b; p.gc = gamma_c; p.btc = beta_c;
  p.sa = state_a; p.sb = state_b; p.sc = state_c;
  p.B = B; p.eps = eps;
  hipStream_t s = (hipStream_t)stream;
  static const int wide = [] { const char* e = getenv("PWR_RESBLOCK_WAVES"); return e ? atoi(e) : 1; }();   // 0: four waves everywhere (round 1)
  if (W == 16 && wide) hipLaunchKernelGGL((resblock_fwd_small_kernel<4, 8>), dim3(B), dim3(512), 0, s, p);
  else if (W == 16) hipLaunchKernelGGL((resblock_fwd_small_kernel<4, 4>), dim3(B), dim3(256), 0, s, p);
  else if (W == 8 && wide) hipLaunchKernelGGL((resblock_fwd_small_kernel<3, 8>), dim3(B), dim3(512), 0, s, p);
  else if (W == 8) hipLaunchKernelGGL((resblock_fwd_small_kernel<3, 4>), dim3(B), dim3(256), 0, s, p);
  else if (W == 4) hipLaunchKernelGGL((resblock_fwd_small_kernel<2, 4>), dim3(B), dim3(256), 0, s, p);
  else hipLaunchKernelGGL((resblock_fwd_small_kernel<1, 4>), dim3(B), dim3(256), 0, s, p);
  return (int)hipGetLastError();
}

extern "C" int pwr_resblock_bwd_small(const void* gout, const void* x, const void* t1, const void* t2, void* dx, void* dt1, void* dt2,
                                      const void* wc_d, const void* wb_d, const void* wa_d, const float* state_a, const float* state_b,
                                      const float* state_c, float* sums_a, float* sums_b, float* sums_c, float* bias_sums, int B, int H,
                                      int W, int C, int dtype, void* stream) {
  if (!(dtype == PWR_BF16 && C == 128 && H == W && (W == 2 || W == 4 || W == 8 || W == 16))) return (int)hipErrorInvalidValue;
  RbBwdParams p;
  p.gout = (const bf16_t*)gout; p.x = (const bf16_t*)x; p.t1 = (const bf16_t*)t1; p.t2 = (const bf16_t*)t2;
  p.dx = (bf16_t*)dx; p.dt1 = (bf16_t*)dt1; p.dt2 = (bf16_t*)dt2;
  p.wcd = (const char*)wc_d; p.wbd = (const char*)wb_d; p.wad = (const char*)wa_d;
  p.sa = state_a; p.sb = state_b; p.sc = state_c;
  p.sums_a = sums_a; p.sums_b = sums_b; p.sums_c = sums_c; p.bias_sums = bias_sums;
  p.B = B;
  hipStream_t s = (hipStream_t)stream;
  static const int wide = [] { const char* e = getenv("PWR_RESBLOCK_WAVES"); return e ? atoi(e) : 1; }();   // 0: four waves everywhere (round 1)
  if (W == 16 && wide) hipLaunchKernelGGL((resblock_bwd_small_kernel<4, 8>), dim3(B), dim3(512), 0, s, p);
  else if (W == 16) hipLaunchKernelGGL((resblock_bwd_small_kernel<4, 4>), dim3(B), dim3(256), 0, s, p);
  else if (W == 8 && wide) hipLaunchKernelGGL((resblock_bwd_small_kernel<3, 8>), dim3(B), dim3(512), 0, s, p);
  else if (W == 8) hipLaunchKernelGGL((resblock_bwd_small_kernel<3, 4>), dim3(B), dim3(256), 0, s, p);
  else if (W == 4) hipLaunchKernelGGL((resblock_bwd_small_kernel<2, 4>), dim3(B), dim3(256), 0, s, p);
  else hipLaunchKernelGGL((resblock_bwd_small_kernel<1, 4>), dim3(B), dim3(256), 0, s, p);
  return (int)hipGetLastError();
}
