// bf16 weight gradient of a 3x3 stride-1 conv with BOTH operands brought into LDS by LDS-DMA (global_load_lds): no register staging,
// no ds_write, no VALU work in the K loop.
//
//   dW[co][ci][ky][kx] = sum_{b, y, x} a[b][y + ky - 1][x + kx - 1][ci] * dy[b][y][x][co]          (/root/reference/model.py: every
//   3x3 Conv2d's weight under autograd; a = the conv's input AFTER its norm + ReLU)
//
// For the layers whose operand needs NO norm on the way (in_norm == NULL: the heads' first convs read the hourglass output f as it
// is, model.py:55 / :104): global_load_lds cannot transform what it moves.  Round 3 also built the other half -- the forward conv
// storing its normalised operand so that EVERY 3x3 layer could take this path -- and dropped it: the extra write cost the chain what
// the side streams gained (6.51 vs 6.48 ms per step; profiles/r3_experiments.md section 5, where the isolated numbers are: 67.9 us
// against 72.0 us for the register-staged conv_wgrad3_kernel incl. the reduce, bit-identical results).
//
// Workgroup = 256 threads (2 x 2 waves), one kernel row ky, a BM (ci) x BN (co) tile, 3 taps kx from one staged input row segment
// with halo (as conv_wgrad3_kernel: same accumulation order, same slabs, same reduce -> bit-identical results).  K step = 32 output
// pixels of one image row.  LDS: a ring of NS stages, each [34 input pixels][BM ch] + [32 pixels][BN ch] as plain rows whose 16-byte
// slots are XOR-swizzled so that the four pixel rows of a ds_read_b64_tr_b16 group fall on four different 64-byte bank quarters
// (cdna_hip_programming.md T10).  A DMA piece is 1 KiB = 64 lanes x 16 B, linear in LDS; the swizzle and the halo go into the per-lane
// SOURCE address.  Out-of-image halo pixels (left / right border) are loaded from a clamped address and zeroed in LDS by the wave
// that issued the piece, behind its own vmcnt wait; an out-of-image input ROW (ky = 0 / 2 at the top / bottom) skips the step's MFMAs.
#include <type_traits>

#include "conv_common.h"
#include "pwr.h"

namespace pwr {

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_w;

// byte offset of 16-byte slot `slot` of row `row` in a tile of RB-byte rows
template <int RB>
__device__ __forceinline__ int wswz(int row, int slot) {
  if constexpr (RB == 256) return row * 256 + ((slot ^ ((row & 3) << 2)) << 4);
  else if constexpr (RB == 128) return row * 128 + ((slot ^ (((row >> 1) & 1) << 2)) << 4);
  else return row * RB + (slot << 4);     // 64-byte rows: four consecutive rows are 256 contiguous bytes
}

// MFMA 32x32x16 operand fragment (8 K values = pixels k0 + 8h .. + 7 of channel chb + lane % 32) through the transposing read.
// INLINE ASM on purpose: behind an LDS-DMA the compiler's wait-count pass puts `s_waitcnt vmcnt(0)` in front of every LDS read it
// knows about (it cannot tell the ring stages apart), i.e. it would wait for the prefetches just issued, every step -- the first
// build of this kernel ran 2.7 us per K step that way.  The reads are invisible to it; wfrag_wait() is the matching lgkmcnt(0).
template <int RB>
__device__ __forceinline__ bf16x8 wfrag(const char* tile, int k0, int chb, int lane) {
  const int li = lane & 15, cg = (lane >> 4) & 1, h = lane >> 5, q = li >> 2, pp = li & 3;
  const int row = k0 + 8 * h + q, slot = (chb >> 3) + 2 * cg + (pp >> 1);
  typedef __attribute__((address_space(3))) const char* lds_cptr;
  const unsigned base = (unsigned)(size_t)(lds_cptr)tile;
  const unsigned a_lo = base + wswz<RB>(row, slot) + 8 * (pp & 1), a_hi = base + wswz<RB>(row + 4, slot) + 8 * (pp & 1);
  bf16x4_w lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a_lo) : "memory");
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(a_hi) : "memory");
  bf16x8 f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3]; f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  return f;
}

template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void conv_wgrad3d_kernel(WgradParams p) {
  typedef bf16_t T;
  typedef bf16x8 V;
  constexpr int KP = 32, NS = 4, XROWS = KP + 2;
  constexpr int XB = BM * 2, YB = BN * 2;                                   // bytes per staged pixel
  constexpr int XCH = (XROWS * XB + 1023) / 1024, YCH = KP * YB / 1024;     // 1-KiB DMA pieces per stage
  constexpr int NCH = XCH + YCH, NCW = (NCH + 3) / 4;                       // every wave issues exactly NCW pieces per step
  constexpr int XBYTES = XCH * 1024, STAGE = NCH * 1024;
  constexpr int MR = BM / 64, NR = BN / 64;
  static_assert(KP * YB % 1024 == 0 && 2 * NCW < 64, "whole pieces, vmcnt range");
  __shared__ __attribute__((aligned(16))) char smem[NS * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  // blockIdx.x enumerates (split, ky) so that the three ky workgroups of one split are 8 ids apart (one XCD, speed only)
  const int grp = blockIdx.x / 24, rr = blockIdx.x - grp * 24;
  const int ky = rr >> 3;
  const int split = grp * 8 + (rr & 7);
  if (split >= p.S) return;
  const int ntn = p.CoutPad / BN;
  const int mtile = blockIdx.y / ntn, ntile = blockIdx.y - mtile * ntn;
  const int ci0 = mtile * BM, co0 = ntile * BN;
  const T* __restrict__ x = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ dy = reinterpret_cast<const T*>(p.dy);
  const int step0 = split * p.steps_per_split;
  const int total_steps = p.M / KP;
  int nsteps = total_steps - step0;
  if (nsteps > p.steps_per_split) nsteps = p.steps_per_split;
  const int tiles_x = p.W / KP;

  // ---- per-lane descriptors of this wave's DMA pieces (constant over the steps; only the tile origin moves).  All selects are
  // arithmetic: a branch around a global_load_lds costs more than the piece
  int d_lds[NCW], d_off[NCW], d_dl[NCW], d_dr[NCW];     // LDS offset in the stage, source offset (elements), clamp deltas (left / right halo)
  bool d_isx[NCW];
#pragma unroll
  for (int i = 0; i < NCW; ++i) {
    int c = wid + 4 * i;
    if (c >= NCH) c = wid;                       // (padding piece: this wave's first piece again, so that every wave counts NCW per step)
    d_lds[i] = c * 1024;
    d_isx[i] = c < XCH;
    const int cx_ = c < XCH ? c : c - XCH;
    const int q = 64 * cx_ + lane;
    // input piece
    const int xr0 = q / (XB / 16), xs1 = q % (XB / 16);
    const int xr = xr0 < XROWS ? xr0 : XROWS - 1;                                    // rows beyond the 34th: nobody reads them
    const int xs = XB == 256 ? (xs1 ^ ((xr0 & 3) << 2)) : (XB == 128 ? (xs1 ^ (((xr0 >> 1) & 1) << 2)) : xs1);
    const int xoff = (xr - 1) * p.Cin + ci0 + 8 * xs;
    // dy piece
    const int yr = q / (YB / 16), ys1 = q % (YB / 16);
    const int ysl = YB == 256 ? (ys1 ^ ((yr & 3) << 2)) : (YB == 128 ? (ys1 ^ (((yr >> 1) & 1) << 2)) : ys1);
    const int yoff = yr * p.Cout + co0 + 8 * ysl;
    d_off[i] = d_isx[i] ? xoff : yoff;
    d_dl[i] = (d_isx[i] && xr == 0) ? p.Cin : 0;             // out-of-image halo: clamped into the row, zeroed in LDS afterwards
    d_dr[i] = (d_isx[i] && xr == XROWS - 1) ? -p.Cin : 0;
  }

  // tile coordinates of the next step to ISSUE and of the next step to CONSUME (incremental, no divisions in the loop)
  int ib, iyy, ix, cb, cy, cx;
  {
    ib = step0 / (p.H * tiles_x);
    const int rem = step0 - ib * p.H * tiles_x;
    iyy = rem / tiles_x; ix = rem - iyy * tiles_x;
    cb = ib; cy = iyy; cx = ix;
  }
  int issued = 0;
  auto issue = [&](int stage) {
    const int iy = iyy + ky - 1;
    const bool rowok = iy >= 0 && iy < p.H;
    const T* xrow = x + (((long long)ib * p.H + (rowok ? iy : iyy)) * p.W + ix * KP) * p.Cin;
    const T* drow = dy + (((long long)ib * p.H + iyy) * p.W + ix * KP) * p.Cout;
    const int first = ix == 0 ? 1 : 0, last = ix == tiles_x - 1 ? 1 : 0;
    char* base = smem + stage * STAGE;
#pragma unroll
    for (int i = 0; i < NCW; ++i) {
      const T* src = (d_isx[i] ? xrow : drow) + (d_off[i] + first * d_dl[i] + last * d_dr[i]);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(base + d_lds[i]), 16, 0, 0);
    }
    // (never past the last tile of this split: further issues re-read it, so that the piece count per step stays constant)
    if (++issued < nsteps) { if (++ix == tiles_x) { ix = 0; if (++iyy == p.H) { iyy = 0; ++ib; } } }
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

  // one K step on ring stage SG (compile-time: every LDS address is base + constant)
  auto body = [&](auto SG, int st) {
    constexpr int stage = decltype(SG)::value;
    char* xs = smem + stage * STAGE;
    char* ys = xs + XBYTES;
    // this wave's pieces of step st have landed (those of the one or two steps issued after it may still be in flight), and every
    // LDS read it issued has retired: a read that is merely issued can still be queued when another wave's DMA lands on the bytes
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    constexpr int W2 = 2 * NCW, W1 = NCW;
    if (st + 2 < nsteps) __builtin_amdgcn_s_waitcnt((W2 & 15) | ((W2 >> 4) << 14) | 0x0070);
    else if (st + 1 < nsteps) __builtin_amdgcn_s_waitcnt((W1 & 15) | ((W1 >> 4) << 14) | 0x0070);
    else __builtin_amdgcn_s_waitcnt(0x0070);
    const int iy = cy + ky - 1;
    const bool rowok = iy >= 0 && iy < p.H;
    if (wid == 0 && lane < XB / 16) {            // wave 0 issued the pieces that hold input pixels 0 and 33 (inline asm: see wfrag)
      typedef __attribute__((address_space(3))) char* lds_ptr;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      const unsigned a0 = (unsigned)(size_t)(lds_ptr)xs + lane * 16, a1 = a0 + (XROWS - 1) * XB;
      if (cx == 0) asm volatile("ds_write_b128 %0, %1" ::"v"(a0), "v"(z) : "memory");
      if (cx == tiles_x - 1) asm volatile("ds_write_b128 %0, %1" ::"v"(a1), "v"(z) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
#ifdef PWR_DEBUG_BUILD
    if (st + 3 < nsteps && !(p.dbg & 4)) issue((stage + 3) & (NS - 1));
    if (rowok && !(p.dbg & 2)) {
#else
    if (st + 3 < nsteps) issue((stage + 3) & (NS - 1));     // into the stage that was read during step st - 1
    if (rowok) {
#endif
      V bf[2][NR], af[2][3][MR];
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
#pragma unroll
        for (int j = 0; j < NR; ++j) bf[ss][j] = wfrag<YB>(ys, ss * 16, wn * NR * 32 + j * 32, lane);
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int i = 0; i < MR; ++i) af[ss][t][i] = wfrag<XB>(xs, ss * 16 + t, wm * MR * 32 + i * 32, lane);
      }
      // all fragment reads of the step are in flight; one wait, tied to the fragments so that no MFMA can be scheduled above it
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
#pragma unroll
        for (int j = 0; j < NR; ++j) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bf[ss][j])::"memory");
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int i = 0; i < MR; ++i) asm volatile("" : "+v"(af[ss][t][i]));
      }
#ifdef PWR_DEBUG_BUILD
      if (!(p.dbg & 1))
#endif
#pragma unroll
      for (int ss = 0; ss < 2; ++ss)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int j = 0; j < NR; ++j)
              acc[t][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ss][t][i], bf[ss][j], acc[t][i][j], 0, 0, 0);
    }
    if (++cx == tiles_x) { cx = 0; if (++cy == p.H) { cy = 0; ++cb; } }
  };

  if (nsteps > 0) {
    issue(0);
    if (nsteps > 1) issue(1);
    if (nsteps > 2) issue(2);
    for (int st = 0; st < nsteps; st += NS) {
      body(std::integral_constant<int, 0>{}, st);
      if (st + 1 < nsteps) body(std::integral_constant<int, 1>{}, st + 1);
      if (st + 2 < nsteps) body(std::integral_constant<int, 2>{}, st + 2);
      if (st + 3 < nsteps) body(std::integral_constant<int, 3>{}, st + 3);
    }
  }
  // (no DMA may still be in flight when the workgroup's LDS is released)
  __builtin_amdgcn_s_waitcnt(0x0070);
  const int r = lane & 31, h = lane >> 5;
#ifdef PWR_DEBUG_BUILD
  if (p.dbg & 8) return;
#endif
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

// The operand is already what the MFMA multiplies (no norm / ReLU to apply), 3x3 stride 1, 32-pixel row segments, 64-channel tiles
bool wgrad3d_applicable(const WgradParams& p) {
  static const bool on = PWR_DBG_ENV("PWR_WGRAD3_DMA", 1) != 0;
  return on && p.in_norm == nullptr && p.ksize == 3 && p.stride == 1 && p.W % 32 == 0 && p.M % 32 == 0 && p.Cin % 64 == 0 && p.Cout % 64 == 0 &&
         p.CoutPad == p.Cout;
}

int launch_wgrad3d(const WgradParams& p0, hipStream_t s) {
  WgradParams p = p0;
  p.dbg = PWR_DBG_ENV("PWR_WGRAD3D_DBG", 0);
  const int bn = p.Cout % 128 == 0 ? 128 : 64;
  dim3 grid(24 * ((p.S + 7) / 8), (p.Cin / 64) * (p.CoutPad / bn), 1), block(256);
  if (bn == 128) hipLaunchKernelGGL((conv_wgrad3d_kernel<64, 128>), grid, block, 0, s, p);
  else hipLaunchKernelGGL((conv_wgrad3d_kernel<64, 64>), grid, block, 0, s, p);
  return (int)hipGetLastError();
}

}  // namespace pwr
