// Implicit-GEMM convolution on the gfx950 matrix cores: forward / data-gradient / weight-gradient.
//
// Replaces every torch.nn.Conv2d with Cin % 4 == 0 of /root/reference/model.py (stem :171-185, stage
// input :137, ResBlock :13-19, heads :54-65 and :103-114) and what autograd derives for them.
//
// Data layout: activations NHWC in T = bf16 or fp32; weights re-packed per (tap, 64-byte
// K-chunk) as rows of 64 bytes so that both MFMA operands are staged as [rows][64 B] LDS tiles with an
// XOR swizzle of the four 16-byte slots (conflict-free ds_read_b128 for the 32x32 MFMA row pattern).
//   bf16: v_mfma_f32_32x32x16_bf16, 32 k per 64-byte row;  fp32: v_mfma_f32_32x32x2_f32 (exact fp32
//   FMA chain, the parity mode), 16 k per row.
// The previous layer's InstanceNorm/BatchNorm + ReLU is applied while the A operand is staged
// ("NR prologue": v = max(0, v*scale[b,c] + shift[b,c])), so a normalised tensor is never written.
// Workgroup = 256 threads = 4 waves; tile = 128 GEMM rows x {32,64,128} columns; fp32 accumulate.
#include "pwr_common.h"
#include "pwr.h"
#include "conv_common.h"
#include <type_traits>

namespace pwr {

// ---------------------------------------------------------------------------------------------
// forward / dgrad
// ---------------------------------------------------------------------------------------------
// PWR_OCC2(cond): second argument of __launch_bounds__ (minimum waves per SIMD) for the kernels meant to run two workgroups per CU.
// Without it hipcc budgets a 256-thread kernel 512 registers and splits accumulators into AGPRs; told "2" it keeps everything within
// 256 VGPRs.  Round 3, the patch conv: 6.62 -> 6.49 ms per train step from this hint alone (profiles/r3_experiments.md section 8).
#ifndef PWR_OCC_HINT
#define PWR_OCC_HINT 1
#endif
#define PWR_OCC2(cond) ((PWR_OCC_HINT && (cond)) ? 2 : 1)

template <typename T, int WM, int WN, int MR, int NR>
__global__ __launch_bounds__(256, PWR_OCC2(sizeof(T) == 2)) void conv_fwd_kernel(ConvParams p) {
  typedef typename Vec16<T>::type V;
  constexpr int KE = Mma<T>::KE, EP = Mma<T>::EP;
  constexpr int BM = WM * MR * 32, BN = WN * NR * 32;
  static_assert(BM == 128, "tile");
  constexpr int NB = (BN * 4 + 255) / 256;       // 16-byte B slots per thread
  constexpr int EROWS = 64, EPITCH = BN + 4;
  constexpr int STAGE_BYTES = 2 * (BM + BN) * 64;
  constexpr int EPI_BYTES = EROWS * EPITCH * 4;
  __shared__ __attribute__((aligned(16))) char smem[STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int mt = xcd_remap(blockIdx.x, gridDim.x), nt = blockIdx.y;
  const int m0 = mt * BM, n0 = nt * BN;
  const T* __restrict__ x = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ w = reinterpret_cast<const T*>(p.w);
  const int HoWo = p.Ho * p.Wo;

  // ---- per-thread A rows: (tid>>2) and (tid>>2)+64, slot q = tid&3.  Everything that does not depend on the (tap, K chunk)
  // of an iteration is computed once: the pointer of tap (0,0), a bit mask of the taps that fall inside the image, the
  // norm-state pointer and the LDS offsets.  Per iteration only wave-uniform offsets are added (scalar ALU).
  const int q = tid & 3;
  const int taps = p.ksize * p.ksize;
  // GEMM row -> (sample, output pixel).  mode 0: NHWC order.  mode 1 (data gradient of a stride-2 conv): a 128-row tile
  // holds output pixels of ONE parity class (oy & 1, ox & 1) and only needs the taps that class can reach (1, 2, 2 or 4 of
  // the 9) -- the other iterations are skipped below, 4x less work.  Consecutive tiles cycle through the four classes so
  // that every XCD's range of tiles carries the same mix of short and long tiles.
  const int Mq = p.M >> 2, HWi = p.H * p.W;
  auto row_pixel = [&](int m, int& b, int& oy, int& ox) -> bool {
    if (p.mode == 0) {
      if (m >= p.M) { b = 0; oy = 0; ox = 0; return false; }
      b = m / HoWo;
      const int rem = m - b * HoWo;
      oy = rem / p.Wo; ox = rem - oy * p.Wo;
      return true;
    }
    const int tile = m / BM, cls = tile & 3, idx = (tile >> 2) * BM + (m - tile * BM);
    if (idx >= Mq) { b = 0; oy = 0; ox = 0; return false; }
    b = idx / HWi;
    const int rem = idx - b * HWi;
    const int yq = rem / p.W;
    oy = 2 * yq + (cls >> 1); ox = 2 * (rem - yq * p.W) + (cls & 1);
    return true;
  };
  const T* xrow[2];
  const float* strow[2];
  unsigned long long vmask[2];    // one bit per tap: k <= 7 -> 49 taps
  int aoff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + (tid >> 2) + 64 * i;
    int b, oy, ox;
    const bool mv = row_pixel(m, b, oy, ox);
    int by, bx;
    unsigned long long mask = 0;
    if (p.mode == 0) {
      by = oy * p.stride - p.pad; bx = ox * p.stride - p.pad;
      for (int ky = 0, t = 0; ky < p.ksize; ++ky)
        for (int kx = 0; kx < p.ksize; ++kx, ++t) {
          const int iy = by + ky, ix = bx + kx;
          if (mv && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) mask |= 1ull << t;
        }
    } else {  // gather form of the stride-2 transposed conv: iy = (oy + pad - ky)/2 when even = by - (ky >> 1)
      by = (oy + p.pad) >> 1; bx = (ox + p.pad) >> 1;
      for (int ky = 0, t = 0; ky < p.ksize; ++ky)
        for (int kx = 0; kx < p.ksize; ++kx, ++t) {
          const int sy = oy + p.pad - ky, sx = ox + p.pad - kx;
          const int iy = sy >> 1, ix = sx >> 1;
          if (mv && !(sy & 1) && !(sx & 1) && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) mask |= 1ull << t;
        }
    }
    vmask[i] = mask;
    xrow[i] = x + ((long long)(b * p.H + by) * p.W + bx) * p.Cin + q * EP;
    strow[i] = p.in_norm ? p.in_norm + (size_t)b * p.Cin + q * EP : nullptr;
    aoff[i] = lds_off((tid >> 2) + 64 * i, q);
  }
  const T* wrow[NB];
  int boff[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int s = tid + 256 * i;
    wrow[i] = w + (size_t)(n0 + (s >> 2)) * KE + (s & 3) * EP;
    boff[i] = BM * 64 + lds_off(s >> 2, s & 3);
  }
  // taps that at least one row of this tile can reach (block-uniform): the K loop runs over those only
  __shared__ unsigned long long s_tapmask;
  if (tid == 0) s_tapmask = 0;
  __syncthreads();
  if (vmask[0] | vmask[1]) atomicOr(&s_tapmask, vmask[0] | vmask[1]);
  __syncthreads();
  const unsigned long long tapmask = s_tapmask;
  const int iters = __popcll(tapmask) * p.KCH;
  const int kmul = (256 + p.ksize - 1) / p.ksize;       // tap / ksize == (tap * kmul) >> 8 for ksize in {1,3,5,7}, tap < ksize^2
  const size_t nplane = (size_t)p.B * p.Cin;
  const long long wstride = (long long)p.CoutPad * KE;      // elements between consecutive (tap, K chunk) weight tiles

  V ra[2], rb[NB];
  bool av[2];
  float nmu[2][EP], nsc[2][EP], nbe[2][EP];

  // (tap, K chunk) of the next tile to load, advanced incrementally: no integer divisions in the K loop
  unsigned long long l_rem = tapmask;
  int l_kch = 0;
  auto load_global = [&](int) {
    const int tap = __ffsll((long long)l_rem) - 1, kch = l_kch;
    const int l_ky = (tap * kmul) >> 8, l_kx = tap - l_ky * p.ksize;
    // wave-uniform element offset of this tap relative to tap (0,0)
    const long long tapoff = p.mode == 0 ? ((long long)l_ky * p.W + l_kx) * p.Cin : -((long long)(l_ky >> 1) * p.W + (l_kx >> 1)) * p.Cin;
    const long long koff = tapoff + kch * KE;
    const bool kfull = kch * KE + q * EP < p.Cin;
    const long long woff = (long long)(tap * p.KCH + kch) * wstride;
    if (++l_kch == p.KCH) { l_kch = 0; l_rem &= l_rem - 1; }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool ok = ((vmask[i] >> tap) & 1ull) && kfull;
      av[i] = ok;
      V v = {};
      if (ok) {
        v = *reinterpret_cast<const V*>(xrow[i] + koff);
        if (p.in_norm) {
          const float* st = strow[i] + kch * KE;
#pragma unroll
          for (int e = 0; e < EP; ++e) { nmu[i][e] = st[e]; nsc[i][e] = st[2 * nplane + e]; nbe[i][e] = st[3 * nplane + e]; }
        }
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int s = tid + 256 * i;
      if (BN * 4 >= 256 * (i + 1) || s < BN * 4) rb[i] = *reinterpret_cast<const V*>(wrow[i] + woff);
    }
  };
  auto store_lds = [&](int buf) {
    char* base = smem + buf * (BM + BN) * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      V v = ra[i];
      if (p.in_norm && av[i]) {
        V o;
#pragma unroll
        for (int e = 0; e < EP; ++e) {
          float f = fmaf(Elem<T>::to_f(v[e]) - nmu[i][e], nsc[i][e], nbe[i][e]);
          if (p.relu_in) f = fmaxf(f, 0.f);
          o[e] = Elem<T>::from_f(f);
        }
        v = o;
      }
      *reinterpret_cast<V*>(base + aoff[i]) = v;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int s = tid + 256 * i;
      if (BN * 4 >= 256 * (i + 1) || s < BN * 4) *reinterpret_cast<V*>(base + boff[i]) = rb[i];
    }
  };

  f32x16 acc[MR][NR];
#pragma unroll
  for (int i = 0; i < MR; ++i)
#pragma unroll
    for (int j = 0; j < NR; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // fp32 (parity) mode: two-level summation -- the MFMA chain is flushed into a second accumulator every 8 K-steps
  // (128 products) so that the rounding error does not grow with the full K = taps*Cin chain.
  constexpr bool kTwoLevel = sizeof(T) == 4;
  f32x16 acc2[kTwoLevel ? MR : 1][kTwoLevel ? NR : 1];
  if constexpr (kTwoLevel) {
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
      for (int j = 0; j < NR; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[i][j][e] = 0.f;
  }
  if (iters > 0) { load_global(0); store_lds(0); }
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    const int buf = it & 1;
    if (it + 1 < iters) load_global(it + 1);
    const char* lA = smem + buf * (BM + BN) * 64;
    mma_tile<T, MR, NR>(lA, lA + BM * 64, wm * MR * 32, wn * NR * 32, lane, acc);
    if constexpr (kTwoLevel) {
      if ((it & 7) == 7 || it + 1 == iters) {
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
          for (int j = 0; j < NR; ++j) { acc2[i][j] += acc[i][j]; acc[i][j] = f32x16{}; }
      }
    }
    if (it + 1 < iters) store_lds(buf ^ 1);
    __syncthreads();
  }
  if constexpr (kTwoLevel) {
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
      for (int j = 0; j < NR; ++j) acc[i][j] = acc2[i][j];
  }

  // ---- epilogue: accumulators -> LDS (fp32, 64 rows at a time) -> coalesced 16-byte stores
  float* E = reinterpret_cast<float*>(smem);
  constexpr int PASSES = BM / EROWS;
  const int r = lane & 31, h = lane >> 5;
  // column statistics (host guarantees mode 0 and Ho*Wo % 128 == 0 when requested: the tile lies in one sample)
  constexpr int CPRS = BN / EP;
  const int tile_b = m0 / HoWo;
  EpiStats<T> est;
  est.init(p, tile_b, n0 + (tid % CPRS) * EP);
#pragma unroll
  for (int ps = 0; ps < PASSES; ++ps) {
    const int wrow0 = wm * MR * 32;  // first tile row of this wave
    if (wrow0 / EROWS == ps) {
      const int er0 = wrow0 - ps * EROWS;
#pragma unroll
      for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int row = er0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            E[row * EPITCH + wn * NR * 32 + j * 32 + r] = acc[i][j][e];
          }
    }
    __syncthreads();
    if (ps == 0) est.set_shift(p, E + (tid % CPRS) * EP, n0 + (tid % CPRS) * EP);
    if (p.y) {
      T* __restrict__ y = reinterpret_cast<T*>(p.y);
      const T* __restrict__ res = reinterpret_cast<const T*>(p.residual);
      constexpr int CPR = BN / EP;               // 16-byte chunks per row
      for (int c = tid; c < EROWS * CPR; c += 256) {
        const int row = c / CPR, cc = (c - row * CPR) * EP;
        const int mg = m0 + ps * EROWS + row, n = n0 + cc;
        int m = mg;
        bool mok = mg < p.M;
        if (p.mode != 0) { int b_, oy_, ox_; mok = row_pixel(mg, b_, oy_, ox_); m = (b_ * p.Ho + oy_) * p.Wo + ox_; }
        if (mok && n < p.Cout) {
          float v[EP];
#pragma unroll
          for (int e = 0; e < EP; ++e) v[e] = E[row * EPITCH + cc + e];
          if (p.bias) {
#pragma unroll
            for (int e = 0; e < EP; ++e) v[e] += p.bias[n + e];
          }
          if (res) {
            V rv = *reinterpret_cast<const V*>(res + (size_t)m * p.Cout + n);
#pragma unroll
            for (int e = 0; e < EP; ++e) v[e] += Elem<T>::to_f(rv[e]);
          }
          V o;
#pragma unroll
          for (int e = 0; e < EP; ++e) o[e] = Elem<T>::from_f(v[e]);
          *reinterpret_cast<V*>(y + (size_t)m * p.Cout + n) = o;
          est.add(p, o, (size_t)m, n);
        }
      }
    }
    if (p.y_nchw) {
      for (int c = tid; c < EROWS * BN; c += 256) {
        const int col = c / EROWS, row = c - col * EROWS;
        const int mg = m0 + ps * EROWS + row, n = n0 + col;
        int m = mg;
        bool mok = mg < p.M;
        if (p.mode != 0) { int b_, oy_, ox_; mok = row_pixel(mg, b_, oy_, ox_); m = (b_ * p.Ho + oy_) * p.Wo + ox_; }
        if (mok && n < p.Cout) {
          float v = E[row * EPITCH + col];
          if (p.bias) v += p.bias[n];
          const int b = m / HoWo, pix = m - b * HoWo;
          p.y_nchw[((size_t)b * p.Cout + n) * HoWo + pix] = v;
        }
      }
    }
    __syncthreads();
  }
  est.template finish<CPRS, 256>(p, E, tile_b, (m0 - tile_b * HoWo) / BM, HoWo / BM, n0);
}

// ---------------------------------------------------------------------------------------------
// weight gradient:  slab[s][tap][ci][co] = sum_{m in split s} NR(x)[m@tap][ci] * dy[m][co]
// GEMM rows = ci (A operand, transposed while staged), cols = co, K = pixels.
// ---------------------------------------------------------------------------------------------
template <typename T, int WM, int WN, int MR, int NR>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradParams p) {
  typedef typename Vec16<T>::type V;
  constexpr int KE = Mma<T>::KE, EP = Mma<T>::EP;   // KE pixels per K step
  constexpr int BM = WM * MR * 32, BN = WN * NR * 32;
  constexpr int ACH = BM / EP;                      // 16-byte channel chunks per pixel (A side)
  constexpr int BCH = BN / EP;
  constexpr int NA = (KE * ACH + 255) / 256, NBL = (KE * BCH + 255) / 256;
  __shared__ __attribute__((aligned(16))) char smem[2 * (BM + BN) * 64];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int tap = blockIdx.x, ky = tap / p.ksize, kx = tap - ky * p.ksize;
  const int ntn = p.CoutPad / BN;
  const int mtile = blockIdx.y / ntn, ntile = blockIdx.y - mtile * ntn;
  const int ci0 = mtile * BM, co0 = ntile * BN;
  const int split = blockIdx.z;
  const T* __restrict__ x = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ dy = reinterpret_cast<const T*>(p.dy);
  const int HoWo = p.Ho * p.Wo;
  const int step0 = split * p.steps_per_split;
  const int total_steps = (p.M + KE - 1) / KE;
  int nsteps = total_steps - step0;
  if (nsteps > p.steps_per_split) nsteps = p.steps_per_split;

  V ra[NA], rb[NBL];
  bool av[NA];
  int abatch[NA];

  auto load_global = [&](int st) {
    const int mbase = (step0 + st) * KE;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int c = tid + 256 * i;
      const int pix = c % KE, cq = c / KE;
      V v = {};
      bool ok = false;
      int b = 0;
      if (c < KE * ACH) {
        const int m = mbase + pix, ci = ci0 + cq * EP;
        if (m < p.M && ci < p.Cin) {
          b = m / HoWo;
          const int rem = m - b * HoWo;
          const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
          const int iy = oy * p.stride + ky - p.pad, ix = ox * p.stride + kx - p.pad;
          if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
            ok = true;
            v = *reinterpret_cast<const V*>(x + ((size_t)(b * p.H + iy) * p.W + ix) * p.Cin + ci);
          }
        }
      }
      ra[i] = v; av[i] = ok; abatch[i] = b;
    }
#pragma unroll
    for (int i = 0; i < NBL; ++i) {
      const int c = tid + 256 * i;
      const int pix = c % KE, cq = c / KE;
      V v = {};
      if (c < KE * BCH) {
        const int m = mbase + pix, co = co0 + cq * EP;
        if (m < p.M && co < p.Cout) v = *reinterpret_cast<const V*>(dy + (size_t)m * p.Cout + co);
      }
      rb[i] = v;
    }
  };
  auto store_lds = [&](int buf) {
    char* lA = smem + buf * (BM + BN) * 64;
    char* lB = lA + BM * 64;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int c = tid + 256 * i;
      if (c < KE * ACH) {
        const int pix = c % KE, cq = c / KE;
        V v = ra[i];
        if (p.in_norm && av[i])
          v = nr_transform<T>(v, p.in_norm + (size_t)abatch[i] * p.Cin + ci0 + cq * EP, (size_t)p.B * p.Cin, p.relu_in);
#pragma unroll
        for (int e = 0; e < EP; ++e) {
          const int row = cq * EP + e;
          *reinterpret_cast<T*>(lA + lds_off(row, pix / EP) + (pix % EP) * sizeof(T)) = v[e];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NBL; ++i) {
      const int c = tid + 256 * i;
      if (c < KE * BCH) {
        const int pix = c % KE, cq = c / KE;
#pragma unroll
        for (int e = 0; e < EP; ++e) {
          const int row = cq * EP + e;
          *reinterpret_cast<T*>(lB + lds_off(row, pix / EP) + (pix % EP) * sizeof(T)) = rb[i][e];
        }
      }
    }
  };

  f32x16 acc[MR][NR];
#pragma unroll
  for (int i = 0; i < MR; ++i)
#pragma unroll
    for (int j = 0; j < NR; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  constexpr bool kTwoLevel = sizeof(T) == 4;
  f32x16 acc2[kTwoLevel ? MR : 1][kTwoLevel ? NR : 1];
  if constexpr (kTwoLevel) {
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
      for (int j = 0; j < NR; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[i][j][e] = 0.f;
  }
  if (nsteps > 0) {
    load_global(0);
    store_lds(0);
    __syncthreads();
    for (int st = 0; st < nsteps; ++st) {
      const int buf = st & 1;
      if (st + 1 < nsteps) load_global(st + 1);
      const char* lA = smem + buf * (BM + BN) * 64;
      mma_tile<T, MR, NR>(lA, lA + BM * 64, wm * MR * 32, wn * NR * 32, lane, acc);
      if constexpr (kTwoLevel) {
        if ((st & 7) == 7 || st + 1 == nsteps) {
#pragma unroll
          for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int j = 0; j < NR; ++j) { acc2[i][j] += acc[i][j]; acc[i][j] = f32x16{}; }
        }
      }
      if (st + 1 < nsteps) store_lds(buf ^ 1);
      __syncthreads();
    }
  }
  if constexpr (kTwoLevel) {
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
      for (int j = 0; j < NR; ++j) acc[i][j] = acc2[i][j];
  }
  // D layout: col = lane&31 -> co (contiguous), row -> ci
  const int r = lane & 31, h = lane >> 5;
  const int taps = p.ksize * p.ksize;
  float* __restrict__ out = p.slab + ((size_t)(split * taps + tap) * p.CinPad) * p.CoutPad;
#pragma unroll
  for (int i = 0; i < MR; ++i)
#pragma unroll
    for (int j = 0; j < NR; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ci = ci0 + wm * MR * 32 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        const int co = co0 + wn * NR * 32 + j * 32 + r;
        out[(size_t)ci * p.CoutPad + co] = acc[i][j][e];
      }
}

// ---------------------------------------------------------------------------------------------
// bf16 weight gradient with hardware-transposed LDS reads.  Both MFMA operands need 8 consecutive K (= pixel)
// values of one channel per lane, while NHWC memory is channel-contiguous.  The tiles are therefore staged row-major
// [pixel][channel] with plain coalesced 16-byte loads / ds_write_b128, and the fragments are read with
// ds_read_b64_tr_b16 (4 pixels x 16 channels per 16-lane group, delivered channel-major).  Row pitch = 2*C + 64 bytes
// keeps the four pixel rows of a group on different banks.
// ---------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
__device__ __forceinline__ bf16x8 frag_tr(const char* tile, int pitch, int k0, int chb, int lane) {
  const int li = lane & 15, cg = (lane >> 4) & 1, h = lane >> 5;
  const char* a0 = tile + (k0 + 8 * h + (li >> 2)) * pitch + (chb + 16 * cg + 4 * (li & 3)) * 2;
  typedef __attribute__((address_space(3))) bf16x4_t* lptr;
  bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lptr)(a0));
  bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lptr)(a0 + 4 * pitch));
  bf16x8 f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3]; f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  return f;
}

template <int WM, int WN, int MR, int NR>
struct WgradTrGeom {
  static constexpr int KP = 32, EP = 8;
  static constexpr int BM = WM * MR * 32, BN = WN * NR * 32;
  static constexpr int PA = BM * 2 + 64, PB = BN * 2 + 64;          // row pitches in bytes
  static constexpr int TILE_A = KP * PA, TILE_B = KP * PB;
  static constexpr int LDS = 2 * (TILE_A + TILE_B);
};

// one workgroup = (tap, output tile `tile_yx`, K split) of one layer
template <int WM, int WN, int MR, int NR>
__device__ __forceinline__ void wgrad_tr_body(const WgradParams& p, int tap, int tile_yx, int split, char* smem) {
  typedef bf16_t T;
  typedef bf16x8 V;
  typedef WgradTrGeom<WM, WN, MR, NR> G;
  constexpr int KP = G::KP, EP = G::EP, BM = G::BM, BN = G::BN, PA = G::PA, PB = G::PB, TILE_A = G::TILE_A, TILE_B = G::TILE_B;
  constexpr int ACH = BM / EP, BCH = BN / EP;                // 16-byte chunks per pixel row
  constexpr int NA = (KP * ACH + 255) / 256, NBL = (KP * BCH + 255) / 256;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int ky = tap / p.ksize, kx = tap - ky * p.ksize;
  const int ntn = p.CoutPad / BN;
  const int mtile = tile_yx / ntn, ntile = tile_yx - mtile * ntn;
  const int ci0 = mtile * BM, co0 = ntile * BN;
  const T* __restrict__ x = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ dy = reinterpret_cast<const T*>(p.dy);
  const int HoWo = p.Ho * p.Wo;
  const int step0 = split * p.steps_per_split;
  const int total_steps = (p.M + KP - 1) / KP;
  int nsteps = total_steps - step0;
  if (nsteps > p.steps_per_split) nsteps = p.steps_per_split;
  const size_t plane = (size_t)p.B * p.Cin;

  V ra[NA], rb[NBL];
  bool av[NA];
  int ab[NA];
  auto load_global = [&](int st) {
    const int mbase = (step0 + st) * KP;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int c = tid + 256 * i;
      const int pix = c / ACH, cq = c % ACH;
      V v = {};
      bool ok = false;
      int b = 0;
      if (c < KP * ACH) {
        const int m = mbase + pix, ci = ci0 + cq * EP;
        if (m < p.M && ci < p.Cin) {
          b = m / HoWo;
          const int rem = m - b * HoWo;
          const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
          const int iy = oy * p.stride + ky - p.pad, ix = ox * p.stride + kx - p.pad;
          if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
            ok = true;
            v = *reinterpret_cast<const V*>(x + ((size_t)(b * p.H + iy) * p.W + ix) * p.Cin + ci);
          }
        }
      }
      ra[i] = v; av[i] = ok; ab[i] = b;
    }
#pragma unroll
    for (int i = 0; i < NBL; ++i) {
      const int c = tid + 256 * i;
      const int pix = c / BCH, cq = c % BCH;
      V v = {};
      if (c < KP * BCH) {
        const int m = mbase + pix, co = co0 + cq * EP;
        if (m < p.M && co < p.Cout) v = *reinterpret_cast<const V*>(dy + (size_t)m * p.Cout + co);
      }
      rb[i] = v;
    }
  };
  auto store_lds = [&](int buf) {
    char* lA = smem + buf * (TILE_A + TILE_B);
    char* lB = lA + TILE_A;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int c = tid + 256 * i;
      if (c < KP * ACH) {
        const int pix = c / ACH, cq = c % ACH;
        V v = ra[i];
        if (p.in_norm && av[i]) v = nr_transform<T>(v, p.in_norm + (size_t)ab[i] * p.Cin + ci0 + cq * EP, plane, p.relu_in);
        *reinterpret_cast<V*>(lA + pix * PA + cq * 16) = v;
      }
    }
#pragma unroll
    for (int i = 0; i < NBL; ++i) {
      const int c = tid + 256 * i;
      if (c < KP * BCH) *reinterpret_cast<V*>(lB + (c / BCH) * PB + (c % BCH) * 16) = rb[i];
    }
  };

  f32x16 acc[MR][NR];
#pragma unroll
  for (int i = 0; i < MR; ++i)
#pragma unroll
    for (int j = 0; j < NR; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  if (nsteps > 0) {
    load_global(0);
    store_lds(0);
    __syncthreads();
    for (int st = 0; st < nsteps; ++st) {
      const int buf = st & 1;
      if (st + 1 < nsteps) load_global(st + 1);
      const char* lA = smem + buf * (TILE_A + TILE_B);
      const char* lB = lA + TILE_A;
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        V a[MR], b[NR];
#pragma unroll
        for (int i = 0; i < MR; ++i) a[i] = frag_tr(lA, PA, ss * 16, wm * MR * 32 + i * 32, lane);
#pragma unroll
        for (int j = 0; j < NR; ++j) b[j] = frag_tr(lB, PB, ss * 16, wn * NR * 32 + j * 32, lane);
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
          for (int j = 0; j < NR; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
      if (st + 1 < nsteps) store_lds(buf ^ 1);
      __syncthreads();
    }
  }
  const int r = lane & 31, h = lane >> 5;
  const int taps = p.ksize * p.ksize;
  float* __restrict__ out = p.slab + ((size_t)(split * taps + tap) * p.CinPad) * p.CoutPad;
#pragma unroll
  for (int i = 0; i < MR; ++i)
#pragma unroll
    for (int j = 0; j < NR; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ci = ci0 + wm * MR * 32 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        const int co = co0 + wn * NR * 32 + j * 32 + r;
        out[(size_t)ci * p.CoutPad + co] = acc[i][j][e];
      }
}

template <int WM, int WN, int MR, int NR>
__global__ __launch_bounds__(256, PWR_OCC2(true)) void conv_wgrad_tr_kernel(WgradParams p) {
  __shared__ __attribute__((aligned(16))) char smem[WgradTrGeom<WM, WN, MR, NR>::LDS];
  wgrad_tr_body<WM, WN, MR, NR>(p, blockIdx.x, blockIdx.y, blockIdx.z, smem);
}

// ---------------------------------------------------------------------------------------------
// GROUPED form: the weight gradients of up to PWR_WGRAD_GROUP_MAX layers in ONE launch.  The 16x16 .. 2x2 maps of the inner
// hourglass (model.py:25-47) carry 8 ResBlocks per stage = 24 conv layers whose weight gradients are microseconds of arithmetic
// each: as 24 launches + 24 split-K reduce launches they were ~16 us apiece of launch latency and tail on the side streams (0.76 ms
// of the 3.4 ms of parameter-gradient work per train step, profiles/r3_serial_kernel_stats.csv).  Workgroup b of the launch looks its
// layer up in a prefix table (wave-uniform), then runs the same body as the single-layer kernel; each layer has its own slab region.
// ---------------------------------------------------------------------------------------------
#define PWR_WGRAD_GROUP_MAX 24
struct WgradGroup {
  WgradParams job[PWR_WGRAD_GROUP_MAX];
  int start[PWR_WGRAD_GROUP_MAX + 1];     // first workgroup of job j; start[n] = grid size
  int n;
};
template <int WM, int WN, int MR, int NR>
__global__ __launch_bounds__(256, PWR_OCC2(true)) void conv_wgrad_tr_group_kernel(WgradGroup g) {
  __shared__ __attribute__((aligned(16))) char smem[WgradTrGeom<WM, WN, MR, NR>::LDS];
  const int b = blockIdx.x;
  int j = 0;
  while (j + 1 < g.n && b >= g.start[j + 1]) ++j;
  const WgradParams& p = g.job[j];
  int r = b - g.start[j];
  const int taps = p.ksize * p.ksize;
  const int tiles = (p.CinPad / WgradTrGeom<WM, WN, MR, NR>::BM) * (p.CoutPad / WgradTrGeom<WM, WN, MR, NR>::BN);
  const int tap = r % taps;
  r /= taps;
  wgrad_tr_body<WM, WN, MR, NR>(p, tap, r % tiles, r / tiles, smem);
}

// ---------------------------------------------------------------------------------------------
// bf16 weight gradient of a 3x3 stride-1 conv, three taps (one kernel row ky) per workgroup.  A K tile is 32
// consecutive output pixels of ONE image row (needs W % 32 == 0); the matching input row segment is staged with a
// one-pixel halo (34 pixels), so the same LDS tile serves kx = 0,1,2 -- the ds_read_b64_tr_b16 row address simply
// shifts by kx.  Per MFMA this loads / normalises / stages 3x less than the one-tap kernel and its integer index math
// is incremental (no divisions in the loop).  192 accumulator registers per wave -> one workgroup per CU; global
// loads are prefetched two tiles ahead in registers.
// ---------------------------------------------------------------------------------------------
// DBG (only instantiated in the debug build, tools/build_debug.py; PWR_WGRAD3_DBG selects): timing by elimination, results are WRONG --
// 1 no MFMAs, 2 no fragment reads, 4 no norm / ReLU math in the staging, 16 no staging stores, 32 no global loads in the loop
// STR = 2 (round 4, second session): the STRIDE-2 3x3 conv (the stem's last layer, model.py:182).  A K tile is still 32 consecutive OUTPUT pixels of
// one output row oy; tap (ky, kx) reads input pixel (2 oy + ky - 1, 2 ox + kx - 1), so the workgroup of kernel row ky stages the 65
// consecutive input pixels 2 ox0 - 1 ... 2 ox0 + 63 of input row 2 oy + ky - 1 (norm + ReLU on the way, column -1 = the conv's zero
// padding) and the fragment of tap kx reads every SECOND staged row starting at row kx -- ds_read_b64_tr_b16 with twice the row pitch.
// The pitch is 2 C + 32 bytes here, so that the four (double-pitch) pixel rows of a 16-lane group still fall on different banks.  The
// layer ran on the one-tap-per-workgroup gather kernel before (296 us alone, 540 us at the tail of the train step for the FLOPs of a
// 44-us head layer: a global round trip and integer divisions per 32-pixel K step).
// KPX = 64 (round 4, second session): a K step of 64 output pixels instead of 32 -- the narrow layers (the heads' 128 -> J conv, the
// 64-channel tiles) spend ~1 us per K step whatever the tile holds (barrier, staging round trip, 6 - 12 MFMAs per wave); half the steps.
// Same pixel order inside a step (four 16-pixel MFMA K blocks instead of two), one staged row segment of 66 pixels; needs W % 64 == 0.
template <int WM, int WN, int MR, int NR, int DEPTH = 2, int DBG = 0, int STR = 1, int KPX = 32>
__global__ __launch_bounds__(256, PWR_OCC2(3 * MR * NR * 16 <= 96)) void conv_wgrad3_kernel(WgradParams p) {
  static_assert(DEPTH >= 2 && DEPTH % 2 == 0, "register stages: even, so that the LDS buffer parity follows the step parity");
  static_assert(STR == 1 || STR == 2, "stride 1 or 2");
  static_assert(KPX == 32 || (KPX == 64 && STR == 1), "K step: 32 pixels, or 64 (stride 1)");
  typedef bf16_t T;
  typedef bf16x8 V;
  constexpr int KP = KPX, EP = 8, AP = STR == 1 ? KP + 2 : 2 * KP + 1;
  constexpr int BM = WM * MR * 32, BN = WN * NR * 32;
  constexpr int PA = BM * 2 + (STR == 1 ? 64 : 32), PB = BN * 2 + 64;
  constexpr int TILE_A = AP * PA, TILE_B = KP * PB;
  constexpr int ACH = BM / EP, BCH = BN / EP;
  constexpr int NA = (AP * ACH + 255) / 256, NBL = (KP * BCH + 255) / 256;
  __shared__ __attribute__((aligned(16))) char smem[2 * (TILE_A + TILE_B)];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  // blockIdx.x enumerates (split, ky) so that the three ky workgroups of one split are 8 ids apart: they land on the
  // same XCD (speed only) and share the dy tiles and the overlapping input rows in that XCD's L2.
  const int grp = blockIdx.x / 24, rr = blockIdx.x - grp * 24;
  const int ky = rr >> 3;
  const int split = grp * 8 + (rr & 7);
  if (split >= p.S) return;
  const int ntn = p.CoutPad / BN;
  const int mtile = blockIdx.y / ntn, ntile = blockIdx.y - mtile * ntn;
  const int ci0 = mtile * BM, co0 = ntile * BN;
  const T* __restrict__ x = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ dy = reinterpret_cast<const T*>(p.dy);
  // (the host counts K steps of 32 pixels; a 64-pixel step covers two of them: ceil(steps / 2) per split still covers everything, and a
  // split beyond the end writes zeros)
  const int sps = KP == 32 ? p.steps_per_split : (p.steps_per_split * 32 + KP - 1) / KP;
  const int step0 = split * sps;
  const int total_steps = p.M / KP;                 // W % KP == 0 -> M % KP == 0
  int nsteps = total_steps - step0;
  if (nsteps > sps) nsteps = sps;
  if (nsteps < 0) nsteps = 0;
  const int OHt = STR == 1 ? p.H : p.Ho;            // rows of the image the K tiles walk (the output)
  const int tiles_x = (STR == 1 ? p.W : p.Wo) / KP;
  const size_t plane = (size_t)p.B * p.Cin;

  // tile coordinates of step `st` (incremental, no divisions in the loop)
  int tb, ty, tx;   // batch, row, x tile of the NEXT tile to be loaded
  {
    // (an empty split -- the 64-pixel re-cut of an odd steps_per_split leaves trailing splits past the end -- starts on the last valid tile:
    // nothing below may ever form an address from tb >= B)
    const int t0 = step0 < total_steps ? step0 : total_steps - 1;
    tb = t0 / (OHt * tiles_x);
    const int rem = t0 - tb * OHt * tiles_x;
    ty = rem / tiles_x; tx = rem - ty * tiles_x;
  }
  // (never past the last tile of this split: the loaders run unconditionally -- a load inside a branch makes the compiler's
  // wait-count pass fall back to vmcnt(0) at the join, which waits for the prefetch that was just issued, every step --
  // and simply re-read the last tile when there is nothing left to fetch)
  int issued = 0;
  auto advance = [&]() {
    if (++issued < nsteps) { if (++tx == tiles_x) { tx = 0; if (++ty == OHt) { ty = 0; ++tb; } } }
  };

  // per-thread NR state for its channel chunk (reloaded when the batch index changes)
  const int acq = tid % ACH;
  float mu[EP], sc[EP], be[EP];
  int state_b = -1;
  auto load_state = [&](int b) {
    if (p.in_norm && b != state_b) {
      const float* stp = p.in_norm + (size_t)b * p.Cin + ci0 + acq * EP;
      if (ci0 + acq * EP < p.Cin) {
#pragma unroll
        for (int e = 0; e < EP; ++e) { mu[e] = stp[e]; sc[e] = stp[2 * plane + e]; be[e] = stp[3 * plane + e]; }
      }
      state_b = b;
    }
  };

  // per-thread constants of the loaders: element offsets relative to the (uniform) tile origin, LDS offsets, edge flags.
  // Loads are branch-free: an out-of-image slot reads a valid address of the same row and is zeroed by a select, so all
  // loads of a stage issue back to back.
  int a_off[NA], a_lds[NA], b_off[NBL], b_lds[NBL];
  bool a_in[NA], a_left[NA], a_right[NA], b_in[NBL];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int c = tid + 256 * i;
    const int pix = c / ACH, cq = c % ACH;     // ACH is a power of two
    const bool chok = ci0 + cq * EP < p.Cin;
    a_in[i] = c < AP * ACH && chok;
    a_left[i] = pix == 0; a_right[i] = STR == 1 && pix == AP - 1;      // (stride 2: the last staged column 2 ox0 + 63 is always inside)
    a_off[i] = a_in[i] ? (pix - 1) * p.Cin + ci0 + cq * EP : 0;
    a_lds[i] = (c < AP * ACH ? pix : 0) * PA + cq * 16;
  }
#pragma unroll
  for (int i = 0; i < NBL; ++i) {
    const int c = tid + 256 * i;
    const int pix = c / BCH, cq = c % BCH;
    b_in[i] = c < KP * BCH && co0 + cq * EP < p.Cout;
    b_off[i] = b_in[i] ? pix * p.Cout + co0 + cq * EP : 0;
    b_lds[i] = (c < KP * BCH ? pix : 0) * PB + cq * 16;
  }
  struct Stage { V a[NA]; V b[NBL]; unsigned okmask; int bidx; bool rowok; };
  Stage sg[DEPTH];   // tiles st+1 .. st+DEPTH-1 in registers (global latency budget: DEPTH-1 steps), tile st in LDS
  auto load_global = [&](Stage& S) {
    const int iy = STR * ty + ky - 1;
    S.rowok = iy >= 0 && iy < p.H;
    S.bidx = tb;
    const T* xrow = x + (((long long)tb * p.H + (S.rowok ? iy : STR * ty)) * p.W + STR * tx * KP) * p.Cin;
    const T* drow = dy + (((long long)tb * OHt + ty) * (tiles_x * KP) + tx * KP) * p.Cout;
    const bool first = tx == 0, last = tx == tiles_x - 1;
    unsigned okm = 0;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const bool ok = a_in[i] && S.rowok && !(a_left[i] && first) && !(a_right[i] && last);
      S.a[i] = *reinterpret_cast<const V*>(xrow + (ok ? a_off[i] : 0));
      okm |= (ok ? 1u : 0u) << i;
    }
    S.okmask = okm;
#pragma unroll
    for (int i = 0; i < NBL; ++i) S.b[i] = *reinterpret_cast<const V*>(drow + b_off[i]);
    advance();
  };
  auto store_lds = [&](Stage& S, int buf) {
    char* lA = smem + buf * (TILE_A + TILE_B);
    char* lB = lA + TILE_A;
    if (DBG & 16) return;
    load_state(S.bidx);
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      V v = S.a[i];
      if (p.in_norm && !(DBG & 4)) {
        V o;
#pragma unroll
        for (int e = 0; e < EP; ++e) {
          float f = fmaf((float)v[e] - mu[e], sc[e], be[e]);
          if (p.relu_in) f = fmaxf(f, 0.f);
          o[e] = (bf16_t)f;
        }
        v = o;
      }
      if (!((S.okmask >> i) & 1)) v = V{};
      if (AP * ACH >= 256 * (i + 1) || tid + 256 * i < AP * ACH) *reinterpret_cast<V*>(lA + a_lds[i]) = v;
    }
#pragma unroll
    for (int i = 0; i < NBL; ++i) {
      V v = S.b[i];
      if (!b_in[i]) v = V{};
      if (KP * BCH >= 256 * (i + 1) || tid + 256 * i < KP * BCH) *reinterpret_cast<V*>(lB + b_lds[i]) = v;
    }
  };

  f32x16 acc[3][MR][NR];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
      for (int j = 0; j < NR; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][i][j][e] = 0.f;

  // one pipeline step with a compile-time stage index (runtime-indexed register arrays would go to scratch)
  auto body = [&](auto KK, int st) {
    constexpr int kk = decltype(KK)::value, buf = kk & 1;
    const bool rowok_cur = sg[kk].rowok;
    if (!(DBG & 32)) load_global(sg[kk]);            // sg[kk] was already stored to LDS: refill with tile st+DEPTH (or the last one again)
    if (rowok_cur) {
      const char* lA = smem + buf * (TILE_A + TILE_B);
      const char* lB = lA + TILE_A;
      // all fragment reads of a 32-pixel half step are issued first (16 fragments, 64 VGPRs): the MFMAs then start as their operands
      // arrive instead of exposing one LDS round trip per group
#pragma unroll
      for (int hh = 0; hh < KP / 32; ++hh) {
        V bf[2][NR], af[2][3][MR];
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
          const int k0 = hh * 32 + ss * 16;
#pragma unroll
          for (int j = 0; j < NR; ++j) bf[ss][j] = (DBG & 2) ? V{(bf16_t)(float)lane} : frag_tr(lB, PB, k0, wn * NR * 32 + j * 32, lane);
#pragma unroll
          for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < MR; ++i)
              af[ss][t][i] = (DBG & 2) ? V{(bf16_t)(float)(lane + t)}
                                       : (STR == 1 ? frag_tr(lA, PA, k0 + t, wm * MR * 32 + i * 32, lane)
                                                   : frag_tr(lA + t * PA, 2 * PA, k0, wm * MR * 32 + i * 32, lane));
        }
#pragma unroll
        for (int ss = 0; ss < 2; ++ss)
#pragma unroll
          for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
              for (int j = 0; j < NR; ++j) {
                if (DBG & 1) asm volatile("" ::"v"(af[ss][t][i]), "v"(bf[ss][j]));
                else acc[t][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ss][t][i], bf[ss][j], acc[t][i][j], 0, 0, 0);
              }
      }
    }
    store_lds(sg[(kk + 1) % DEPTH], buf ^ 1);        // (after the last step: a tile nobody reads)
    __syncthreads();
  };
  if (nsteps > 0) {
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) load_global(sg[k]);
    store_lds(sg[0], 0);
    __syncthreads();
    for (int st = 0; st < nsteps; st += DEPTH) {
      body(std::integral_constant<int, 0>{}, st);
      if (st + 1 < nsteps) body(std::integral_constant<int, 1>{}, st + 1);
      if constexpr (DEPTH == 4) {
        if (st + 2 < nsteps) body(std::integral_constant<int, 2>{}, st + 2);
        if (st + 3 < nsteps) body(std::integral_constant<int, 3>{}, st + 3);
      }
    }
  }
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    float* __restrict__ out = p.slab + ((size_t)(split * 9 + ky * 3 + t) * p.CinPad) * p.CoutPad;
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
      for (int j = 0; j < NR; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int ci = ci0 + wm * MR * 32 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          const int co = co0 + wn * NR * 32 + j * 32 + r;
          out[(size_t)ci * p.CoutPad + co] = acc[t][i][j][e];
        }
  }
}

// dW[co][ci][ky][kx] (+)= sum_s slab[s][tap][ci][co]   (fixed order -> deterministic): one block per (32 co, 1 ci) row,
// thread = (float4 of co, slab group g of 32), slabs k = g, g + 32, ... ascending, ALL of a thread's slab loads of a round (SL slabs x
// TAPS items) in flight before the first add; the 8 groups of a wave are combined with shuffles, the 4 waves through LDS in a fixed
// order, and the OIHW rows are written in runs of `taps` floats.  (Round 1's form issued one slab's items, waited, added, and went
// round again: 2.5 dependent rounds of DRAM latency on 256 blocks = 25 us for the 47 MB of a 128 -> 128 3x3 layer, 1.9 TB/s; this
// form sums in the same order -- bit-identical, checked in round 2 -- and takes 13 us.)
template <int TC, int SL>
__device__ __forceinline__ void wgrad_reduce_fast_body(const float* __restrict__ slab, float* __restrict__ dw, int S, int taps, int Cout,
                                                       int CinPad, int CoutPad, int cin_real, int accumulate, int bx, int by, int bz,
                                                       float* tile) {
  // bz: chunk of TC taps (9 taps = one chunk; 25 / 49 taps of a 5x5 / 7x7 layer = 3 / 6 chunks)
  constexpr int pitch = TC + 1;
  const int co0 = bx * 32, ci = by, tap0 = bz * TC;
  const int nt = taps - tap0 < TC ? taps - tap0 : TC;
  const int c4 = threadIdx.x & 7, grp = threadIdx.x >> 3, wave = threadIdx.x >> 6;
  const int co = co0 + c4 * 4;
  f32x4 acc[TC];
#pragma unroll
  for (int t = 0; t < TC; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const size_t stride = (size_t)taps * CinPad * CoutPad, tstride = (size_t)CinPad * CoutPad;
  const float* base = slab + (size_t)ci * CoutPad + co + tap0 * tstride;
  for (int k0 = grp; k0 < S; k0 += 32 * SL) {
    f32x4 v[SL][TC];
#pragma unroll
    for (int u = 0; u < SL; ++u) {
      const int k = k0 + 32 * u;
      const float* q = base + (size_t)(k < S ? k : grp) * stride;     // (unconditional loads; the add below is predicated)
#pragma unroll
      for (int t = 0; t < TC; ++t) v[u][t] = *reinterpret_cast<const f32x4*>(q + (t < nt ? t : 0) * tstride);
    }
#pragma unroll
    for (int u = 0; u < SL; ++u) {
      if (k0 + 32 * u < S) {
#pragma unroll
        for (int t = 0; t < TC; ++t) acc[t] += v[u][t];
      }
    }
  }
#pragma unroll
  for (int t = 0; t < TC; ++t) {
    // (lane ^ 8, 16, 32 by DPP / v_permlane swaps, pwr_common.h: the sums of the __shfl_xor loop without its ds_bpermute_b32 round trips)
    acc[t].x = lane_xor_add<8>(acc[t].x); acc[t].y = lane_xor_add<8>(acc[t].y); acc[t].z = lane_xor_add<8>(acc[t].z); acc[t].w = lane_xor_add<8>(acc[t].w);
    acc[t].x = lane_xor_add<16>(acc[t].x); acc[t].y = lane_xor_add<16>(acc[t].y); acc[t].z = lane_xor_add<16>(acc[t].z); acc[t].w = lane_xor_add<16>(acc[t].w);
    acc[t].x = lane_xor_add<32>(acc[t].x); acc[t].y = lane_xor_add<32>(acc[t].y); acc[t].z = lane_xor_add<32>(acc[t].z); acc[t].w = lane_xor_add<32>(acc[t].w);
    if ((threadIdx.x & 63) < 8) {
      float* q = tile + (wave * 32 + c4 * 4) * pitch + t;
      q[0] = acc[t].x; q[pitch] = acc[t].y; q[2 * pitch] = acc[t].z; q[3 * pitch] = acc[t].w;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 32 * TC; i += 256) {
    const int col2 = i / TC, rem = i - col2 * TC;
    const int co2 = co0 + col2;
    if (co2 < Cout && rem < nt) {
      float r = 0.f;
#pragma unroll
      for (int g = 0; g < 4; ++g) r += tile[(g * 32 + col2) * pitch + rem];
      const size_t o = ((size_t)co2 * cin_real + ci) * taps + tap0 + rem;
      dw[o] = accumulate ? dw[o] + r : r;
    }
  }
}
template <int TC, int SL>
__global__ __launch_bounds__(256) void wgrad_reduce_fast_kernel(const float* __restrict__ slab, float* __restrict__ dw, int S, int taps, int Cout,
                                                                int CinPad, int CoutPad, int cin_real, int accumulate) {
  __shared__ float tile[4 * 32 * (TC + 1)];
  wgrad_reduce_fast_body<TC, SL>(slab, dw, S, taps, Cout, CinPad, CoutPad, cin_real, accumulate, blockIdx.x, blockIdx.y, blockIdx.z, tile);
}

// grouped form (see conv_wgrad_tr_group_kernel): block b -> (job, 32-co tile, ci); <= 9 taps per layer
struct ReduceJob { const float* slab; float* dw; int S, taps, Cout, CinPad, CoutPad, cin_real, accumulate, start; };
struct ReduceGroup { ReduceJob job[PWR_WGRAD_GROUP_MAX]; int n, total; };
template <int TC, int SL>
__global__ __launch_bounds__(256) void wgrad_reduce_group_kernel(ReduceGroup g) {
  __shared__ float tile[4 * 32 * (TC + 1)];
  const int b = blockIdx.x;
  int j = 0;
  while (j + 1 < g.n && b >= g.job[j + 1].start) ++j;
  const ReduceJob& q = g.job[j];
  const int r = b - q.start, ncb = (q.Cout + 31) / 32;
  wgrad_reduce_fast_body<TC, SL>(q.slab, q.dw, q.S, q.taps, q.Cout, q.CinPad, q.CoutPad, q.cin_real, q.accumulate, r % ncb, r / ncb, 0, tile);
}

// ---------------------------------------------------------------------------------------------
// weight packing: OIHW fp32 -> [tap][kch][RowsPad][KE] T
//   kind 0 (forward): rows = cout, k = cin            value = W[row][k][ky][kx]
//   kind 1 (dgrad of a stride-1 conv): rows = cin, k = cout, taps flipped   value = W[k][row][K-1-ky][K-1-kx]
//   kind 2 (dgrad of a stride-2 conv, gather form): rows = cin, k = cout    value = W[k][row][ky][kx]
// order 1 (bf16, 3x3, 128 rows x 128 k only): the same 16-byte units in conv_wstat.hip's FRAGMENT order -- [tap][kch][wave = row / 32][ss][lane]
//   with lane = r + 32 h holding unit (row = 32 wave + chan(r), k slot 2 ss + h), chan(r) = 16 (r / 4 % 2) + 4 (r / 8) + r % 4: every weight
//   fragment of that kernel is then ONE contiguous 1-KiB load (from the standard order it is 32 pieces of 32 B per wave instruction, and
//   the 72 such loads per lane were 13 000 of a workgroup's 63 000 cycles).  Callers pass such a pack with bit 0 of its address set.
// ---------------------------------------------------------------------------------------------
struct PackDesc {
  long long src_off;   // floats into the flat parameter buffer
  long long dst_off;   // bytes into the pack buffer
  int Cout, Cin, ksize, kind, rows_pad, KCH, dtype, order;
};

// One thread = 8 consecutive K elements of one pack row: 8 gathered floats (stride = taps for the forward packs, Cin x taps for the
// data-gradient packs), ONE 16-byte store (bf16; two for fp32), 32-bit index arithmetic (round 4: the element-per-thread form with
// 64-bit divisions and 2-byte stores took 37 us at the head of every train step for 13 MB of parameters).
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ flat, char* __restrict__ packs,
                                                           const PackDesc* __restrict__ descs) {
  const PackDesc d = descs[blockIdx.y];
  const int KE = d.dtype == PWR_BF16 ? 32 : 16, KE8 = KE / 8;
  const int taps = d.ksize * d.ksize;
  const int total8 = taps * d.KCH * d.rows_pad * KE8;
  const float* W = flat + d.src_off;
  const int rows = d.kind == 0 ? d.Cout : d.Cin;
  const int kdim = d.kind == 0 ? d.Cin : d.Cout;
  // element stride between consecutive k of one row in the OIHW source
  const int kstride = d.kind == 0 ? taps : d.Cin * taps;
  for (int i8 = blockIdx.x * 256 + threadIdx.x; i8 < total8; i8 += gridDim.x * 256) {
    const int e0 = (i8 % KE8) * 8;
    int r = i8 / KE8;
    const int row = r % d.rows_pad; r /= d.rows_pad;
    const int kch = r % d.KCH;
    const int tap = r / d.KCH;
    const int k0 = kch * KE + e0;
    int ky = tap / d.ksize, kx = tap - ky * d.ksize;
    if (d.kind == 1) { ky = d.ksize - 1 - ky; kx = d.ksize - 1 - kx; }
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    if (row < rows) {
      // W[co][ci][ky][kx]: kind 0: co = row, ci = k; kinds 1 / 2: co = k, ci = row
      const float* src = d.kind == 0 ? W + ((size_t)row * d.Cin * taps + ky * d.ksize + kx) : W + ((size_t)row * taps + ky * d.ksize + kx);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (k0 + j < kdim) v[j] = src[(size_t)(k0 + j) * kstride];
    }
    const size_t i = (size_t)i8 * 8;
    if (d.dtype == PWR_BF16) {
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (bf16_t)v[j];
      size_t unit = i8;
      // the descriptors are device-resident (the host cannot check them at this call): order 1 is honoured only for the one shape it is
      // defined for -- kind 0, 128 x 128, 3x3, bf16 (rows_pad 128, KCH 4) -- and any other record falls back to the standard order, inside
      // its own allocation
      if (d.order == 1 && d.kind == 0 && d.rows_pad == 128 && d.KCH == 4 && d.ksize == 3 && d.Cout == 128 && d.Cin == 128) {
        const int slot = i8 % KE8, wave = row >> 5, c = row & 31;
        const int r = (c & 3) | (((c >> 4) & 1) << 2) | (((c >> 2) & 3) << 3);       // inverse of chan(r)
        unit = (size_t)(tap * d.KCH + kch) * 512 + wave * 128 + (slot >> 1) * 64 + r + 32 * (slot & 1);
      }
      *reinterpret_cast<bf16x8*>(packs + d.dst_off + unit * 16) = o;
    } else {
      float* dst = reinterpret_cast<float*>(packs + d.dst_off) + i;
      *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
      *reinterpret_cast<f32x4*>(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
  }
}


template <typename T>
static int launch_conv(const ConvParams& p, hipStream_t s) {
  const int bn = pick_bn(p.Cout);
  if (p.CoutPad % bn) return PWR_EINVAL;
  if (p.w_frag && p.mode != 0) return PWR_EINVAL;      // (no kernel below the patch dispatch reads a fragment-order pack)
  if (conv_tr2_applicable(p, sizeof(T) == 2 ? PWR_BF16 : PWR_F32)) return launch_conv_tr2(p, s);
  if (conv_patch_applicable(p, sizeof(T) == 2 ? PWR_BF16 : PWR_F32)) return launch_conv_patch(p, sizeof(T) == 2 ? PWR_BF16 : PWR_F32, s);
  if (p.w_frag) return PWR_EINVAL;
  // mode 1: four parity classes of M/4 rows each, every class padded to whole tiles
  const int mtiles = p.mode == 0 ? (p.M + 127) / 128 : 4 * ((p.M / 4 + 127) / 128);
  dim3 grid(mtiles, p.CoutPad / bn), block(256);
  if (bn == 128) hipLaunchKernelGGL((conv_fwd_kernel<T, 2, 2, 2, 2>), grid, block, 0, s, p);
  else if (bn == 64) hipLaunchKernelGGL((conv_fwd_kernel<T, 2, 2, 2, 1>), grid, block, 0, s, p);
  else hipLaunchKernelGGL((conv_fwd_kernel<T, 4, 1, 1, 1>), grid, block, 0, s, p);
  return (int)hipGetLastError();
}

template <typename T>
static int launch_wgrad(const WgradParams& p, hipStream_t s) {
  const int bn = pick_bn(p.Cout);
  if (p.CoutPad % bn || p.CinPad % 128) return PWR_EINVAL;
  const int taps = p.ksize * p.ksize;
  dim3 grid(taps, (p.CinPad / 128) * (p.CoutPad / bn), p.S), block(256);
  if constexpr (sizeof(T) == 2) {
#ifdef PWR_DEBUG_BUILD
    if (wgrad9w_applicable(p)) return launch_wgrad9w(p, s);                  // (debug build, PWR_WGRAD9W=1: the nine-tap experiment, measured slower)
#endif
    if (wgrad3w_applicable(p)) return launch_wgrad3w(p, nullptr, s);         // whole 128-channel tiles: the wave-specialised three-tap kernel (the product path)
    // K steps of 64 pixels for the narrow three-tap layers on maps whose width is a multiple of 64 (round 4; PWR_WGRAD3_KP64 bits, debug
    // build: 1 = <= 32 output channels -- the heads' 128 -> J conv --, 2 = 64 x 64 tiles, 4 = the 64 x 128 tile of <= 64 input channels,
    // 8 = also the norm-fed 64 -> 64 layers that the LDS-DMA kernel would take).  Isolated at the engine's splits: 61.5 -> 46.4 us,
    // 95 -> 78, 175 -> 141, 51.3 -> 37.9; train step 5.523 -> 5.464 ms with all four (5.505 / 5.494 / 5.482 with bits 1 / 6 / 9).
    static const int kp64 = PWR_DBG_ENV("PWR_WGRAD3_KP64", 15);
    const bool w64 = p.ksize == 3 && p.stride == 1 && p.W % 64 == 0 && p.M % 64 == 0;
    if (w64 && (kp64 & 8) && bn == 64 && p.Cin <= 64) {
      dim3 g64(24 * ((p.S + 7) / 8), ((p.Cin + 63) / 64) * (p.CoutPad / bn), 1);
      hipLaunchKernelGGL((conv_wgrad3_kernel<2, 2, 1, 1, 2, 0, 1, 64>), g64, block, 0, s, p);
      return (int)hipGetLastError();
    }
    if (wgrad3d_applicable(p)) return launch_wgrad3d(p, s);                  // operands by LDS-DMA (no norm to apply on the way)
    if (p.ksize == 3 && p.stride == 2 && p.Wo % 32 == 0 && p.H == 2 * p.Ho && p.W == 2 * p.Wo && bn == 128 && p.Cin % 64 == 0 &&
        PWR_DBG_ENV("PWR_WGRAD3_S2", 1)) {
      // the stride-2 form of the three-tap kernel (64 (ci) x 128 (co) tile, two workgroups per CU)
      dim3 g64(24 * ((p.S + 7) / 8), (p.Cin / 64) * (p.CoutPad / bn), 1);
      hipLaunchKernelGGL((conv_wgrad3_kernel<2, 2, 1, 2, 2, 0, 2>), g64, block, 0, s, p);
      return (int)hipGetLastError();
    }
    if (p.ksize == 3 && p.stride == 1 && p.W % 32 == 0 && p.M % 32 == 0) {   // three taps per workgroup
      dim3 g3(24 * ((p.S + 7) / 8), grid.y, 1);
      // 128 output channels: 64 (ci) x 128 (co) x 3 taps per workgroup = 96 accumulator registers -> TWO workgroups per CU,
      // so one's norm/ReLU staging (VALU) and waits overlap the other's MFMAs; the x tile (the operand that needs VALU work)
      // is split between them, not duplicated.  (128 x 128 x 3 = 192 registers allows one wave per SIMD only: 25 % MFMA busy.)
      if (w64 && (kp64 & 1) && bn == 32) { hipLaunchKernelGGL((conv_wgrad3_kernel<4, 1, 1, 1, 2, 0, 1, 64>), g3, block, 0, s, p); return (int)hipGetLastError(); }
      if (w64 && (kp64 & 2) && bn == 64 && p.Cin <= 64) {
        dim3 g64(g3.x, ((p.Cin + 63) / 64) * (p.CoutPad / bn), 1);
        hipLaunchKernelGGL((conv_wgrad3_kernel<2, 2, 1, 1, 2, 0, 1, 64>), g64, block, 0, s, p);
        return (int)hipGetLastError();
      }
      if (w64 && (kp64 & 4) && bn == 128 && p.Cin <= 64) {
        dim3 g64(g3.x, ((p.Cin + 63) / 64) * (p.CoutPad / bn), 1);
        hipLaunchKernelGGL((conv_wgrad3_kernel<2, 2, 1, 2, 2, 0, 1, 64>), g64, block, 0, s, p);
        return (int)hipGetLastError();
      }
      static const bool bm64 = (PWR_DBG_ENV("PWR_WGRAD3_BM64", 1) != 0);
#ifdef PWR_DEBUG_BUILD
      if (bn == 128 && p.Cin <= 64 && PWR_DBG_ENV("PWR_WGRAD3_CO64", 0)) {
        // (experiment, debug build: the stem's 64 -> 128 conv at 128 x 128 as two 64 x 64 tiles per split = two workgroups per CU instead
        // of one 64 x 128 tile = one per CU at the engine's 80 splits: 184 - 194 us against 176 - 184, no gain -- the loop is bound by the
        // per-step round trip, not by the workgroups per CU)
        dim3 g64(g3.x, p.CoutPad / 64, 1);
        if (PWR_DBG_ENV("PWR_WGRAD3_DEPTH", 2) == 4) hipLaunchKernelGGL((conv_wgrad3_kernel<2, 2, 1, 1, 4>), g64, block, 0, s, p);
        else hipLaunchKernelGGL((conv_wgrad3_kernel<2, 2, 1, 1>), g64, block, 0, s, p);
        return (int)hipGetLastError();
      }
#endif
      if (bn == 128 && bm64) {
        dim3 g64(g3.x, ((p.Cin + 63) / 64) * (p.CoutPad / bn), 1);
#ifdef PWR_DEBUG_BUILD
        // (experiment: the 64 x 128 tile with 64-pixel K steps for the 128-channel layers as well -- reached with PWR_WGRAD3W=0)
        if (w64 && (kp64 & 16)) { hipLaunchKernelGGL((conv_wgrad3_kernel<2, 2, 1, 2, 2, 0, 1, 64>), g64, block, 0, s, p); return (int)hipGetLastError(); }
        // (experiments, debug build only: register prefetch depth 4 -- measured 82.6 vs 81.7 us, not latency-bound -- and the
        // timing-by-elimination variants)
        static const int depth = PWR_DBG_ENV("PWR_WGRAD3_DEPTH", 2);
        if (depth == 4) { hipLaunchKernelGGL((conv_wgrad3_kernel<2, 2, 1, 2, 4>), g64, block, 0, s, p); return (int)hipGetLastError(); }
        static const int dbg = PWR_DBG_ENV("PWR_WGRAD3_DBG", 0);
        switch (dbg) {
          case 1: hipLaunchKernelGGL((conv_wgrad3_kernel<2, 2, 1, 2, 2, 1>), g64, block, 0, s, p); return (int)hipGetLastError();
          case 2: hipLaunchKernelGGL((conv_wgrad3_kernel<2, 2, 1, 2, 2, 2>), g64, block, 0, s, p); return (int)hipGetLastError();
          case 3: hipLaunchKernelGGL((conv_wgrad3_kernel<2, 2, 1, 2, 2, 3>), g64, block, 0, s, p); return (int)hipGetLastError();
          case 4: hipLaunchKernelGGL((conv_wgrad3_kernel<2, 2, 1, 2, 2, 4>), g64, block, 0, s, p); return (int)hipGetLastError();
          case 16: hipLaunchKernelGGL((conv_wgrad3_kernel<2, 2, 1, 2, 2, 16>), g64, block, 0, s, p); return (int)hipGetLastError();
          case 19: hipLaunchKernelGGL((conv_wgrad3_kernel<2, 2, 1, 2, 2, 19>), g64, block, 0, s, p); return (int)hipGetLastError();
          case 48: hipLaunchKernelGGL((conv_wgrad3_kernel<2, 2, 1, 2, 2, 48>), g64, block, 0, s, p); return (int)hipGetLastError();
          case 51: hipLaunchKernelGGL((conv_wgrad3_kernel<2, 2, 1, 2, 2, 51>), g64, block, 0, s, p); return (int)hipGetLastError();
          default: break;
        }
#endif
        hipLaunchKernelGGL((conv_wgrad3_kernel<2, 2, 1, 2>), g64, block, 0, s, p);
        return (int)hipGetLastError();
      }
      if (bn == 128) hipLaunchKernelGGL((conv_wgrad3_kernel<2, 2, 2, 2>), g3, block, 0, s, p);
      else if (bn == 64 && p.Cin <= 64 && PWR_DBG_ENV("PWR_WGRAD3_BM64N64", 1)) {
        // <= 64 input channels (the stem's 32 -> 64 conv, model.py:171): a 64 x 64 tile -- the 128-row tile multiplied 96 rows of zeros
        dim3 g64(g3.x, ((p.Cin + 63) / 64) * (p.CoutPad / bn), 1);
#ifdef PWR_DEBUG_BUILD
        if (PWR_DBG_ENV("PWR_WGRAD3_DEPTH", 2) == 4) { hipLaunchKernelGGL((conv_wgrad3_kernel<2, 2, 1, 1, 4>), g64, block, 0, s, p); return (int)hipGetLastError(); }
#endif
        hipLaunchKernelGGL((conv_wgrad3_kernel<2, 2, 1, 1>), g64, block, 0, s, p);
      }
      else if (bn == 64) hipLaunchKernelGGL((conv_wgrad3_kernel<2, 2, 2, 1>), g3, block, 0, s, p);
      else hipLaunchKernelGGL((conv_wgrad3_kernel<4, 1, 1, 1>), g3, block, 0, s, p);
      return (int)hipGetLastError();
    }
    if (bn == 128) hipLaunchKernelGGL((conv_wgrad_tr_kernel<2, 2, 2, 2>), grid, block, 0, s, p);
    else if (bn == 64) hipLaunchKernelGGL((conv_wgrad_tr_kernel<2, 2, 2, 1>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((conv_wgrad_tr_kernel<4, 1, 1, 1>), grid, block, 0, s, p);
  } else {
    if (bn == 128) hipLaunchKernelGGL((conv_wgrad_kernel<T, 2, 2, 2, 2>), grid, block, 0, s, p);
    else if (bn == 64) hipLaunchKernelGGL((conv_wgrad_kernel<T, 2, 2, 2, 1>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((conv_wgrad_kernel<T, 4, 1, 1, 1>), grid, block, 0, s, p);
  }
  return (int)hipGetLastError();
}

}  // namespace pwr

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" int pwr_conv_out_pad(int cout) { int bn = pwr::pick_bn(cout); return (cout + bn - 1) / bn * bn; }

extern "C" size_t pwr_conv_pack_bytes(int cout, int cin, int ksize, int kind, int dtype) {
  const int KE = dtype == PWR_BF16 ? 32 : 16, esz = dtype == PWR_BF16 ? 2 : 4;
  const int rows = kind == 0 ? cout : cin, kdim = kind == 0 ? cin : cout;
  const int rows_pad = pwr_conv_out_pad(rows), KCH = (kdim + KE - 1) / KE;
  return (size_t)ksize * ksize * KCH * rows_pad * KE * esz;
}

extern "C" int pwr_pack_weights(const float* flat_params, void* packs, const void* descs_dev, int n_desc, void* stream) {
  if (n_desc <= 0) return 0;
  hipLaunchKernelGGL(pwr::pack_weights_kernel, dim3(32, n_desc), dim3(256), 0, (hipStream_t)stream, flat_params,
                     (char*)packs, (const pwr::PackDesc*)descs_dev);
  return (int)hipGetLastError();
}

static int conv_params_fill(pwr::ConvParams& p, const void* x, const void* wpack, const float* bias, const float* in_norm, int relu_in,
                            const void* residual, void* y, float* y_nchw, int B, int H, int W, int Cin, int Cout, int ksize,
                            int stride, int mode, int dtype) {
  const int EP = dtype == PWR_BF16 ? 8 : 4, KE = dtype == PWR_BF16 ? 32 : 16;
  if (Cin % EP || (y && Cout % EP) || (ksize != 1 && ksize != 3 && ksize != 5 && ksize != 7) || (stride != 1 && stride != 2)) return PWR_EUNSUPPORTED;
  p.x = x; p.bias = bias; p.in_norm = in_norm; p.residual = residual;
  p.w_frag = (int)((uintptr_t)wpack & 1);                       // (a fragment-order pack: only the shapes conv_wstat.hip takes may carry one)
  p.w = reinterpret_cast<const void*>((uintptr_t)wpack & ~(uintptr_t)1);
  // (conv_wstat_shape prices mode 0 only: a tagged pack on the transposed mode would reach launch_conv_tr2 / the universal kernel, which
  // read the standard order)
  if (p.w_frag && (mode != 0 || !pwr::conv_wstat_shape(B, H, W, Cin, Cout, ksize, stride, dtype))) return PWR_EINVAL;
  p.y = y; p.y_nchw = y_nchw; p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
  p.ksize = ksize; p.pad = ksize / 2; p.mode = mode; p.relu_in = relu_in;
  if (mode == 0) {
    p.stride = stride;
    p.Ho = (H + 2 * p.pad - ksize) / stride + 1; p.Wo = (W + 2 * p.pad - ksize) / stride + 1;
  } else {  // transposed stride-2: x is dy [B,H,W,Cin], output is [B,2H,2W,Cout] (even sizes, pad = k/2)
    p.stride = 1; p.Ho = 2 * H; p.Wo = 2 * W;
  }
  p.CoutPad = pwr_conv_out_pad(Cout);
  p.KCH = (Cin + KE - 1) / KE;
  p.M = B * p.Ho * p.Wo;
  return 0;
}

extern "C" int pwr_conv_fwd(const void* x, const void* wpack, const float* bias, const float* in_norm, int relu_in,
                            const void* residual, void* y, float* y_nchw, int B, int H, int W, int Cin, int Cout, int ksize,
                            int stride, int mode, int dtype, void* stream) {
  pwr::ConvParams p;
  const int rc = conv_params_fill(p, x, wpack, bias, in_norm, relu_in, residual, y, y_nchw, B, H, W, Cin, Cout, ksize, stride, mode, dtype);
  if (rc) return rc;
  return dtype == PWR_BF16 ? pwr::launch_conv<bf16_t>(p, (hipStream_t)stream) : pwr::launch_conv<float>(p, (hipStream_t)stream);
}

// Debugging aid: while set, every 3x3 patch-conv workgroup writes 8 int64 (s_memtime at start / patch loaded / patch staged /
// K loop done / end, -, HW_ID, XCC_ID) to stamps[(blockIdx.y * gridDim.x + blockIdx.x) * 8].  NULL switches it off.
#ifdef PWR_DEBUG_BUILD
extern "C" void pwr_debug_set_stamps(void* stamps) { pwr::set_debug_stamps((long long*)stamps); }
extern "C" void pwr_debug_set_delay(int d) { pwr::set_debug_delay(d); }
#endif
// Debugging aid: 1 / 0 forces the ping-pong form of the 3x3 128->128 conv on / off (where it applies), -1 = the default (env PWR_PINGPONG)

// slab rows per sample that pwr_conv_fwd_stats writes for this conv shape; 0 = the shape cannot produce column statistics
// (a 128-pixel tile would straddle samples, or the transposed mode)
extern "C" int pwr_conv_stats_chunks(int H, int W, int Cin, int Cout, int ksize, int stride, int mode, int dtype) {
  pwr::ConvParams p;
  if (conv_params_fill(p, nullptr, nullptr, nullptr, nullptr, 0, nullptr, (void*)1, nullptr, 1, H, W, Cin, Cout, ksize, stride, mode, dtype)) return 0;
  static const bool on = (PWR_DBG_ENV("PWR_CONV_STATS", 1) != 0);
  if (!on) return 0;
  // mode 1 (the data gradient of a stride-2 conv, H x W = the gradient's map): the four parity classes of the patch kernel (one launch) write the
  // norm-backward sums of the tensor they produce (round 4; nb_partial only)
  if (mode == 1) return (pwr::conv_tr2_applicable(p, dtype) && PWR_DBG_ENV("PWR_TR2_STATS", 1)) ? pwr::conv_tr2_stats_chunks(p, dtype) : 0;
  if (mode != 0) return 0;
  if (pwr::conv_patch_applicable(p, dtype)) return pwr::conv_patch_stats_chunks(p, dtype);
  const int HoWo = p.Ho * p.Wo;
  return HoWo % 128 == 0 ? HoWo / 128 : 0;
}

// pwr_conv_fwd + per-channel column statistics of the output in the epilogue (exactly one of st_partial / nb_partial):
//   st_partial: [B*chunks][2][Cout] shifted sums of the stored output (shift = bias, or 0): forward statistics of the norm
//               that follows (model.py conv -> norm pairs), finished by pwr_norm_finalize_partial;
//   nb_partial: this launch is a data gradient producing g; sums of relu-masked g and g*xhat w.r.t. (nb_y, nb_state): the
//               reductions of the norm backward, finished by pwr_norm_bwd_from_partial.
extern "C" int pwr_conv_fwd_stats(const void* x, const void* wpack, const float* bias, const float* in_norm, int relu_in,
                                  const void* residual, void* y, int B, int H, int W, int Cin, int Cout, int ksize, int stride,
                                  int mode, float* st_partial, const void* nb_y, const float* nb_state, float* nb_partial,
                                  int nb_relu, int dtype, void* stream) {
  pwr::ConvParams p;
  const int rc = conv_params_fill(p, x, wpack, bias, in_norm, relu_in, residual, y, nullptr, B, H, W, Cin, Cout, ksize, stride, mode, dtype);
  if (rc) return rc;
  if ((st_partial != nullptr) == (nb_partial != nullptr) || !y) return PWR_EINVAL;
  if (mode == 1 && st_partial) return PWR_EUNSUPPORTED;
  if (pwr_conv_stats_chunks(H, W, Cin, Cout, ksize, stride, mode, dtype) == 0) return PWR_EUNSUPPORTED;
  p.st_partial = st_partial;
  p.nb_y = nb_y; p.nb_state = nb_state; p.nb_partial = nb_partial; p.nb_relu = nb_relu;
  return dtype == PWR_BF16 ? pwr::launch_conv<bf16_t>(p, (hipStream_t)stream) : pwr::launch_conv<float>(p, (hipStream_t)stream);
}

// Two pwr_conv_fwd_stats launches (forward statistics form, stride 1, no residual) of ONE shape as one launch; PWR_EUNSUPPORTED when the
// shape has no pair kernel (the caller then launches them one after the other)
extern "C" int pwr_conv_fwd_stats_pair(const void* xa, const void* wa, const float* bias_a, const float* in_norm_a, void* ya, float* st_partial_a,
                                       const void* xb, const void* wb, const float* bias_b, const float* in_norm_b, void* yb, float* st_partial_b,
                                       int relu_in, int B, int H, int W, int Cin, int Cout, int ksize, int dtype, void* stream) {
  pwr::ConvParams a, b;
  int rc = conv_params_fill(a, xa, wa, bias_a, in_norm_a, relu_in, nullptr, ya, nullptr, B, H, W, Cin, Cout, ksize, 1, 0, dtype);
  if (rc) return rc;
  rc = conv_params_fill(b, xb, wb, bias_b, in_norm_b, relu_in, nullptr, yb, nullptr, B, H, W, Cin, Cout, ksize, 1, 0, dtype);
  if (rc) return rc;
  if (!st_partial_a || !st_partial_b || !ya || !yb) return PWR_EINVAL;
  if (dtype != PWR_BF16 || !pwr::conv_patch_pair_applicable(a, b, dtype)) return PWR_EUNSUPPORTED;
  a.st_partial = st_partial_a; b.st_partial = st_partial_b;
  return pwr::launch_conv_patch_pair(a, b, (hipStream_t)stream);
}

// Two pwr_conv_fwd launches with fp32 NCHW outputs only (the two heads' last convs, model.py:64 / :113) of ONE shape as one launch of
// the narrow weight-stationary kernel; PWR_EUNSUPPORTED when the shape has no such launch (the caller then launches them one after the other)
extern "C" int pwr_conv_fwd_nchw_pair(const void* xa, const void* wa, const float* bias_a, const float* in_norm_a, float* ya_nchw,
                                      const void* xb, const void* wb, const float* bias_b, const float* in_norm_b, float* yb_nchw,
                                      int relu_in, int B, int H, int W, int Cin, int Cout, int ksize, int dtype, void* stream) {
  pwr::ConvParams a, b;
  if (!ya_nchw || !yb_nchw) return PWR_EINVAL;
  int rc = conv_params_fill(a, xa, wa, bias_a, in_norm_a, relu_in, nullptr, nullptr, ya_nchw, B, H, W, Cin, Cout, ksize, 1, 0, dtype);
  if (rc) return rc;
  rc = conv_params_fill(b, xb, wb, bias_b, in_norm_b, relu_in, nullptr, nullptr, yb_nchw, B, H, W, Cin, Cout, ksize, 1, 0, dtype);
  if (rc) return rc;
  if (dtype != PWR_BF16 || !pwr::conv_wstat_narrow_pair_applicable(a, b, dtype)) return PWR_EUNSUPPORTED;
  return pwr::launch_conv_wstat(a, &b, (hipStream_t)stream);
}

// Two data gradients of stride-1 3x3 convs of ONE shape (pwr_conv_fwd_stats in its norm-backward form: x = dy, kind-1 pack, no bias, no
// prologue; nb_partial = the reductions of the norm backward of the tensor the gradient belongs to) as one launch: the two regression
// heads walk their three 128 -> 128 convs backwards in lock-step (model.py:54-65 / :103-114)
extern "C" int pwr_conv_dgrad_stats_pair(const void* dya, const void* wa, void* dxa, const void* nb_y_a, const float* nb_state_a, float* nb_partial_a,
                                         const void* dyb, const void* wb, void* dxb, const void* nb_y_b, const float* nb_state_b, float* nb_partial_b,
                                         int nb_relu, int B, int H, int W, int Cin, int Cout, int ksize, int dtype, void* stream) {
  pwr::ConvParams a, b;
  int rc = conv_params_fill(a, dya, wa, nullptr, nullptr, 0, nullptr, dxa, nullptr, B, H, W, Cin, Cout, ksize, 1, 0, dtype);
  if (rc) return rc;
  rc = conv_params_fill(b, dyb, wb, nullptr, nullptr, 0, nullptr, dxb, nullptr, B, H, W, Cin, Cout, ksize, 1, 0, dtype);
  if (rc) return rc;
  if (!nb_partial_a || !nb_partial_b || !dxa || !dxb || !nb_y_a || !nb_y_b || !nb_state_a || !nb_state_b) return PWR_EINVAL;
  if (dtype != PWR_BF16 || !pwr::conv_patch_pair_applicable(a, b, dtype)) return PWR_EUNSUPPORTED;
  a.nb_y = nb_y_a; a.nb_state = nb_state_a; a.nb_partial = nb_partial_a; a.nb_relu = nb_relu;
  b.nb_y = nb_y_b; b.nb_state = nb_state_b; b.nb_partial = nb_partial_b; b.nb_relu = nb_relu;
  return pwr::launch_conv_patch_pair(a, b, (hipStream_t)stream);
}

// pwr_conv_dgrad_stats_pair whose inputs are the RAW gradients g = dL/d relu(norm(fb_y)) of the layer above: the norm backward that stood
// between two data gradients (pwr_norm_bwd_apply_from_partial) runs in this launch's staging, from the slab the data gradient above wrote
// (ConvParams::fb_*; conv_patch.hip, FB).  fb_dy_* receive the dy that launch would have written (the weight gradients' operand).
extern "C" int pwr_conv_dgrad_fold_stats_pair(const void* ga, const void* wa, void* dxa, const void* nb_y_a, const float* nb_state_a, float* nb_partial_a,
                                              const void* fb_y_a, const float* fb_state_a, const float* fb_partial_a, void* fb_dy_a,
                                              const void* gb, const void* wb, void* dxb, const void* nb_y_b, const float* nb_state_b, float* nb_partial_b,
                                              const void* fb_y_b, const float* fb_state_b, const float* fb_partial_b, void* fb_dy_b,
                                              int fb_pchunks, int fb_relu, int nb_relu, int B, int H, int W, int Cin, int Cout, int ksize, int dtype,
                                              void* stream) {
  pwr::ConvParams a, b;
  int rc = conv_params_fill(a, ga, wa, nullptr, nullptr, 0, nullptr, dxa, nullptr, B, H, W, Cin, Cout, ksize, 1, 0, dtype);
  if (rc) return rc;
  rc = conv_params_fill(b, gb, wb, nullptr, nullptr, 0, nullptr, dxb, nullptr, B, H, W, Cin, Cout, ksize, 1, 0, dtype);
  if (rc) return rc;
  if (!nb_partial_a || !nb_partial_b || !dxa || !dxb || !nb_y_a || !nb_y_b || !nb_state_a || !nb_state_b) return PWR_EINVAL;
  if (!fb_y_a || !fb_y_b || !fb_state_a || !fb_state_b || !fb_partial_a || !fb_partial_b || !fb_dy_a || !fb_dy_b || fb_pchunks < 1) return PWR_EINVAL;
  if (fb_dy_a == ga || fb_dy_b == gb) return PWR_EINVAL;       // (a tile's halo reads its neighbours' RAW pixels: dy cannot replace g in place)
  if (dtype != PWR_BF16 || Cin != 128 || ksize != 3 || !pwr::conv_patch_pair_applicable(a, b, dtype)) return PWR_EUNSUPPORTED;
  a.nb_y = nb_y_a; a.nb_state = nb_state_a; a.nb_partial = nb_partial_a; a.nb_relu = nb_relu;
  b.nb_y = nb_y_b; b.nb_state = nb_state_b; b.nb_partial = nb_partial_b; b.nb_relu = nb_relu;
  a.fb_y = fb_y_a; a.fb_state = fb_state_a; a.fb_partial = fb_partial_a; a.fb_dy = fb_dy_a; a.fb_pchunks = fb_pchunks; a.fb_relu = fb_relu;
  b.fb_y = fb_y_b; b.fb_state = fb_state_b; b.fb_partial = fb_partial_b; b.fb_dy = fb_dy_b; b.fb_pchunks = fb_pchunks; b.fb_relu = fb_relu;
  return pwr::launch_conv_patch_pair(a, b, (hipStream_t)stream);
}

extern "C" size_t pwr_conv_wgrad_slab_bytes(int cout, int cin, int ksize, int splits) {
  const int cinpad = (cin + 127) / 128 * 128;
  return (size_t)splits * ksize * ksize * cinpad * pwr_conv_out_pad(cout) * sizeof(float);
}

extern "C" int pwr_conv_wgrad(const void* x, const void* dy, const float* in_norm, int relu_in,
                              float* slab, float* dw, int accumulate, int B, int H, int W, int Cin, int cin_real, int Cout,
                              int cout_real, int ksize, int stride, int splits, int dtype, void* stream) {
  const int EP = dtype == PWR_BF16 ? 8 : 4, KE = dtype == PWR_BF16 ? 32 : 16;
  if (Cin % EP || Cout % EP || (ksize != 1 && ksize != 3 && ksize != 5 && ksize != 7) || splits < 1) return PWR_EUNSUPPORTED;
  pwr::WgradParams p;
  p.x = x; p.dy = dy; p.in_norm = in_norm; p.slab = slab;
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.ksize = ksize; p.stride = stride; p.pad = ksize / 2;
  p.Ho = (H + 2 * p.pad - ksize) / stride + 1; p.Wo = (W + 2 * p.pad - ksize) / stride + 1;
  p.CoutPad = pwr_conv_out_pad(Cout); p.CinPad = (Cin + 127) / 128 * 128;
  p.relu_in = relu_in; p.M = B * p.Ho * p.Wo;
  const int total_steps = (p.M + KE - 1) / KE;
  p.steps_per_split = (total_steps + splits - 1) / splits;
  p.S = (total_steps + p.steps_per_split - 1) / p.steps_per_split;   // effective splits (<= requested)
  hipStream_t s = (hipStream_t)stream;
  int rc = dtype == PWR_BF16 ? pwr::launch_wgrad<bf16_t>(p, s) : pwr::launch_wgrad<float>(p, s);
  if (rc) return rc;
  if (cout_real <= 0 || cout_real > Cout || cin_real <= 0 || cin_real > Cin) return PWR_EINVAL;
  const int taps = ksize * ksize;
  const dim3 g((cout_real + 31) / 32, cin_real, (taps + 8) / 9);
  if (taps > 1)
    hipLaunchKernelGGL((pwr::wgrad_reduce_fast_kernel<9, 3>), g, dim3(256), 0, s, slab, dw, p.S, taps, cout_real, p.CinPad, p.CoutPad, cin_real, accumulate);
  else
    hipLaunchKernelGGL((pwr::wgrad_reduce_fast_kernel<1, 16>), g, dim3(256), 0, s, slab, dw, p.S, taps, cout_real, p.CinPad, p.CoutPad, cin_real, accumulate);
  return (int)hipGetLastError();
}

static void wgrad_params_fill(pwr::WgradParams& p, const void* x, const void* dy, const float* in_norm, int relu_in, float* slab, int B, int H, int W,
                              int Cin, int Cout, int ksize, int stride, int splits, int KE) {
  p.x = x; p.dy = dy; p.in_norm = in_norm; p.slab = slab;
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.ksize = ksize; p.stride = stride; p.pad = ksize / 2;
  p.Ho = (H + 2 * p.pad - ksize) / stride + 1; p.Wo = (W + 2 * p.pad - ksize) / stride + 1;
  p.CoutPad = pwr_conv_out_pad(Cout); p.CinPad = (Cin + 127) / 128 * 128;
  p.relu_in = relu_in; p.M = B * p.Ho * p.Wo;
  const int total_steps = (p.M + KE - 1) / KE;
  p.steps_per_split = (total_steps + splits - 1) / splits;
  p.S = (total_steps + p.steps_per_split - 1) / p.steps_per_split;   // effective splits (<= requested)
}

// Two pwr_conv_wgrad calls of ONE geometry (3x3, stride 1, bf16, whole 128-channel tiles, no padded channels) as ONE launch of the
// wave-specialised kernel (conv_wgrad_ws.hip) + ONE reduce launch: the two regression heads' convs of the same depth.  `slab` holds both
// jobs' slabs (2 x pwr_conv_wgrad_slab_bytes).  Results are those of two single calls with the same `splits`, bit for bit.
extern "C" int pwr_conv_wgrad_pair(const void* xa, const void* dya, const float* in_norm_a, float* dwa, const void* xb, const void* dyb,
                                   const float* in_norm_b, float* dwb, int relu_in, float* slab, int B, int H, int W, int Cin, int Cout,
                                   int splits, int dtype, void* stream) {
  if (dtype != PWR_BF16 || splits < 1 || Cin % 128 || Cout % 128) return PWR_EUNSUPPORTED;
  pwr::WgradParams a, b;
  wgrad_params_fill(a, xa, dya, in_norm_a, relu_in, slab, B, H, W, Cin, Cout, 3, 1, splits, 32);
  const size_t half = pwr_conv_wgrad_slab_bytes(Cout, Cin, 3, splits) / sizeof(float);
  wgrad_params_fill(b, xb, dyb, in_norm_b, relu_in, slab + half, B, H, W, Cin, Cout, 3, 1, splits, 32);
  if (!pwr::wgrad3w_applicable(a) || (in_norm_a == nullptr) != (in_norm_b == nullptr)) return PWR_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  int rc = pwr::launch_wgrad3w(a, &b, s);
  if (rc) return rc;
  pwr::ReduceGroup r;
  const int per = ((Cout + 31) / 32) * Cin;
  r.n = 2; r.total = 2 * per;
  r.job[0] = pwr::ReduceJob{slab, dwa, a.S, 9, Cout, a.CinPad, a.CoutPad, Cin, 0, 0};
  r.job[1] = pwr::ReduceJob{slab + half, dwb, b.S, 9, Cout, b.CinPad, b.CoutPad, Cin, 0, per};
  hipLaunchKernelGGL((pwr::wgrad_reduce_group_kernel<9, 3>), dim3(r.total), dim3(256), 0, s, r);
  return (int)hipGetLastError();
}


// ---- grouped weight gradients (bf16, stride 1, ksize 1 or 3): see conv_wgrad_tr_group_kernel
namespace {
struct GroupPlan { pwr::WgradParams p; int bn, S, blocks; size_t slab_off; };
// K splits of one layer inside a grouped launch: ~16 K steps (512 pixels) per workgroup, at most 32 splits
static int group_splits(int M) {
  const int steps = (M + 31) / 32;
  int s = (steps + 15) / 16;
  if (s > 32) s = 32;
  return s < 1 ? 1 : s;
}
static int plan_group(const pwr_wgrad_job* jobs, int njobs, int B, GroupPlan* out, size_t* slab_bytes) {
  size_t off = 0;
  for (int j = 0; j < njobs; ++j) {
    const pwr_wgrad_job& q = jobs[j];
    if (q.Cin % 8 || q.Cout % 8 || (q.ksize != 1 && q.ksize != 3) || q.cin_real <= 0 || q.cin_real > q.Cin || q.cout_real <= 0 || q.cout_real > q.Cout)
      return PWR_EUNSUPPORTED;
    pwr::WgradParams& p = out[j].p;
    p.x = q.x; p.dy = q.dy; p.in_norm = q.in_norm; p.slab = nullptr;
    p.B = B; p.H = q.H; p.W = q.W; p.Cin = q.Cin; p.Cout = q.Cout; p.ksize = q.ksize; p.stride = 1; p.pad = q.ksize / 2;
    p.Ho = q.H; p.Wo = q.W;
    p.CoutPad = pwr_conv_out_pad(q.Cout); p.CinPad = (q.Cin + 127) / 128 * 128;
    p.relu_in = q.relu_in; p.M = B * q.H * q.W;
    const int total_steps = (p.M + 31) / 32, want = group_splits(p.M);
    p.steps_per_split = (total_steps + want - 1) / want;
    p.S = (total_steps + p.steps_per_split - 1) / p.steps_per_split;
    out[j].bn = pwr::pick_bn(q.Cout);
    out[j].S = p.S;
    out[j].blocks = q.ksize * q.ksize * (p.CinPad / 128) * (p.CoutPad / out[j].bn) * p.S;
    out[j].slab_off = off;
    off += (size_t)p.S * q.ksize * q.ksize * p.CinPad * p.CoutPad * sizeof(float);
  }
  *slab_bytes = off;
  return 0;
}
}  // namespace

extern "C" size_t pwr_conv_wgrad_group_slab_bytes(const pwr_wgrad_job* jobs, int njobs, int B) {
  if (njobs < 1 || njobs > 2 * PWR_WGRAD_GROUP_MAX) return 0;
  GroupPlan plan[2 * PWR_WGRAD_GROUP_MAX];
  size_t bytes = 0;
  if (plan_group(jobs, njobs, B, plan, &bytes)) return 0;
  return bytes;
}

extern "C" int pwr_conv_wgrad_group(const pwr_wgrad_job* jobs, int njobs, float* slab, int B, int dtype, void* stream) {
  if (dtype != PWR_BF16) return PWR_EUNSUPPORTED;
  if (njobs < 1 || njobs > 2 * PWR_WGRAD_GROUP_MAX) return PWR_EINVAL;
  GroupPlan plan[2 * PWR_WGRAD_GROUP_MAX];
  size_t bytes = 0;
  int rc = plan_group(jobs, njobs, B, plan, &bytes);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  // one launch per N-tile class (the kernels' register tiles are template parameters)
  for (int bn = 32; bn <= 128; bn *= 2) {
    pwr::WgradGroup g;
    g.n = 0;
    int total = 0;
    for (int j = 0; j < njobs; ++j) {
      if (plan[j].bn != bn) continue;
      if (g.n == PWR_WGRAD_GROUP_MAX) return PWR_EINVAL;
      g.job[g.n] = plan[j].p;
      g.job[g.n].slab = reinterpret_cast<float*>(reinterpret_cast<char*>(slab) + plan[j].slab_off);
      g.start[g.n++] = total;
      total += plan[j].blocks;
    }
    if (!g.n) continue;
    g.start[g.n] = total;
    if (bn == 128) hipLaunchKernelGGL((pwr::conv_wgrad_tr_group_kernel<2, 2, 2, 2>), dim3(total), dim3(256), 0, s, g);
    else if (bn == 64) hipLaunchKernelGGL((pwr::conv_wgrad_tr_group_kernel<2, 2, 2, 1>), dim3(total), dim3(256), 0, s, g);
    else hipLaunchKernelGGL((pwr::conv_wgrad_tr_group_kernel<4, 1, 1, 1>), dim3(total), dim3(256), 0, s, g);
  }
  // split-K reduce of all layers: one launch for the 3x3 layers, one for the 1x1 layers
  for (int ks = 1; ks <= 3; ks += 2) {
    pwr::ReduceGroup r;
    r.n = 0;
    int total = 0;
    for (int j = 0; j < njobs; ++j) {
      if (jobs[j].ksize != ks) continue;
      while (r.n == PWR_WGRAD_GROUP_MAX) {      // (more layers of one kind than a launch takes: flush)
        r.total = total;
        if (ks == 3) hipLaunchKernelGGL((pwr::wgrad_reduce_group_kernel<9, 3>), dim3(total), dim3(256), 0, s, r);
        else hipLaunchKernelGGL((pwr::wgrad_reduce_group_kernel<1, 16>), dim3(total), dim3(256), 0, s, r);
        r.n = 0; total = 0;
      }
      const pwr::WgradParams& p = plan[j].p;
      r.job[r.n++] = pwr::ReduceJob{reinterpret_cast<const float*>(reinterpret_cast<const char*>(slab) + plan[j].slab_off), jobs[j].dw, p.S, ks * ks,
                                    jobs[j].cout_real, p.CinPad, p.CoutPad, jobs[j].cin_real, 0, total};
      total += ((jobs[j].cout_real + 31) / 32) * jobs[j].cin_real;
    }
    if (!r.n) continue;
    r.total = total;
    if (ks == 3) hipLaunchKernelGGL((pwr::wgrad_reduce_group_kernel<9, 3>), dim3(total), dim3(256), 0, s, r);
    else hipLaunchKernelGGL((pwr::wgrad_reduce_group_kernel<1, 16>), dim3(total), dim3(256), 0, s, r);
  }
  return (int)hipGetLastError();
}
