// bf16 weight gradient of a 3x3 stride-1 conv with BOTH operands brought into LDS by LDS-DMA (global_load_lds): no register staging,
// no per-step address arithmetic; the K loop is waits, one barrier, DMA issues, transposing fragment reads and MFMAs.
//
//   dW[co][ci][ky][kx] = sum_{b, y, x} a[b][y + ky - 1][x + kx - 1][ci] * dy[b][y][x][co]          (/root/reference/model.py: every
//   3x3 Conv2d's weight under autograd; a = the conv's input AFTER its norm + ReLU)
//
// global_load_lds cannot transform what it moves, so there are two forms:
//   NRM = false  the operand needs NO norm on the way (in_norm == NULL: the heads' first convs read the hourglass output f as it is,
//                model.py:55 / :104).  Bit-identical to the register-staged conv_wgrad3_kernel, 57 against 70 us at the C2 heads shape.
//   NRM = true   the landed raw input tile of step st + 1 is read back from LDS, normalised + ReLU'd with conv_wgrad3_kernel's
//                arithmetic (bit-identical again) and stored in place while step st's MFMAs issue.  Pays for 64-wide output tiles
//                (48 -> 27 us per launch in the step's serial profile), not for 128-wide ones (86 against 76 us: 78 VALU instructions
//                per wave and step that the MFMAs of the bigger tile do not hide) -- wgrad3d_applicable() picks accordingly.
//   (Round 3 also built the third way -- the forward conv storing its normalised operand -- and dropped it: the extra write cost the
//   chain what the side streams gained, profiles/r3_experiments.md section 5.)
//
// Workgroup = 256 threads (2 x 2 waves), one kernel row ky, a 64 (ci) x BN (co) tile, 3 taps kx from one staged input row segment
// with halo (as conv_wgrad3_kernel: same accumulation order, same slabs, same reduce).  K step = 32 output pixels of one image row.
// LDS: a ring of NS = 4 (NRM: 5) stages, each [34 input pixels][64 ch] + [32 pixels][BN ch] as plain rows whose 16-byte slots are
// XOR-swizzled so that the four pixel rows of a ds_read_b64_tr_b16 group fall on four different 64-byte bank quarters
// (cdna_hip_programming.md T10).  A DMA piece is 1 KiB = 64 lanes x 16 B, linear in LDS; the swizzle and the halo go into the per-lane
// SOURCE address.  Out-of-image halo pixels (left / right border) are loaded from a clamped address and zeroed in LDS (wave 0 behind
// its own vmcnt wait; NRM: by the norm pass); an out-of-image input ROW (ky = 0 / 2 at the top / bottom) skips the step's MFMAs.
//
// LDS reads and the norm pass's stores are INLINE ASM: the compiler's wait-count pass would put vmcnt(0) -- a wait for every prefetch
// in flight -- in front of any LDS access it knows about behind an LDS-DMA.  The price: the compiler believes an asm's output is
// defined when the statement ends, while the data arrives later.  Rules kept here: one definition and one tied wait per asynchronous
// register, straight-line code between a read and its wait, and pixelwiseregression_amd/codeobj_scan.py::async_lds_hazards checks the
// linked code object for any instruction that touches such a register too early (profiles/r3_experiments.md section 14); the gate runs
// in build() with the LLVM tools of the hipcc in use and FAILS the build where they are missing, its result is recorded beside the library
// and tests/test_boundary_cpu.py refuses a library without a clean record (round 4).  Since round 4 the 128-channel layers run on
// conv_wgrad_ws.hip instead, whose MFMA waves read LDS through compiler-visible loads and whose loader waves' reads are each tied to
// their wait; this kernel remains for the 64-channel tiles.
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
// build of this kernel ran 2.7 us per K step that way.  The reads are invisible to it; the caller's lgkmcnt waits are tied to the
// fragment registers.  The lane-dependent part of the address (wfrag_lane: swizzled row / slot of the lane, for k0 < 16) lives in one
// VGPR per (tile, tap) for the whole kernel; ring stage, K half and the +4 rows of the second read go into the instruction's
// offset field (the swizzles repeat every 4 rows, so they are plain row offsets).
template <int RB>
__device__ __forceinline__ unsigned wfrag_lane(int k0, int chb, int lane) {
  const int li = lane & 15, cg = (lane >> 4) & 1, h = lane >> 5, q = li >> 2, pp = li & 3;
  return wswz<RB>(k0 + 8 * h + q, (chb >> 3) + 2 * cg + (pp >> 1)) + 8 * (pp & 1);
}
template <int RB, int OFF>
__device__ __forceinline__ bf16x8 wfrag(unsigned lane_addr) {
  static_assert(OFF >= 0 && OFF + 4 * RB < 65536, "ds offset field");
  bf16x4_w lo, hi;
  // (early-clobber outputs: the destination may not share a register with the address, which the second read still needs)
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=&v"(lo) : "v"(lane_addr), "n"(OFF) : "memory");
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=&v"(hi) : "v"(lane_addr), "n"(OFF + 4 * RB) : "memory");
  bf16x8 f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3]; f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  return f;
}

template <int BM, int BN, bool NRM>
__global__ __launch_bounds__(256, 2) void conv_wgrad3d_kernel(WgradParams p) {
  typedef bf16_t T;
  typedef bf16x8 V;
  constexpr int KP = 32, XROWS = KP + 2;
  constexpr int NS = NRM ? 5 : 4, D = NS - 1;            // ring stages; a step's pieces are issued D steps ahead (NRM: one of them goes to the norm pass)
  constexpr int XB = BM * 2, YB = BN * 2;                                   // bytes per staged pixel
  constexpr int XCH = (XROWS * XB + 1023) / 1024, YCH = KP * YB / 1024;     // 1-KiB DMA pieces per stage
  constexpr int NCH = XCH + YCH, NCW = (NCH + 3) / 4;                       // every wave issues exactly NCW pieces per step
  constexpr int XBYTES = XCH * 1024, STAGE = NCH * 1024;
  constexpr int MR = BM / 64, NR = BN / 64;
  static_assert(KP * YB % 1024 == 0 && 3 * NCW < 64, "whole pieces, vmcnt range");
  static_assert(!NRM || XB == 128, "the in-LDS norm pass is laid out for 64-channel input tiles");
  constexpr int MAXSB = 4;                                                   // NRM: norm states of at most this many samples per split
  __shared__ __attribute__((aligned(16))) char smem[NS * STAGE + (NRM ? MAXSB * 3 * BM * 4 : 0)];

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);      // (wave-uniform: scalar registers)
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
  // (the halo pixels 0 and 33 sit in pieces 0 and I_R * 4, both issued by wave 0 as its pieces 0 and I_R: one clamp delta each)
  constexpr int I_R = ((XROWS - 1) * XB / 1024) / 4;
  static_assert(((XROWS - 1) * XB / 1024) % 4 == 0 && I_R != 0, "the right halo pixel is in one of wave 0's pieces");
  int d_lds[NCW], d_off[NCW], d_dl = 0, d_dr = 0;       // LDS offset in the stage, source offset (elements), clamp deltas (left / right halo)
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
    if (i == 0) d_dl = (d_isx[i] && xr == 0) ? p.Cin : 0;            // out-of-image halo: clamped into the row, zeroed in LDS afterwards
    if (i == I_R) d_dr = (d_isx[i] && xr == XROWS - 1) ? -p.Cin : 0;
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
  auto issue = [&](int soff) {               // soff: byte offset of the ring stage
    const int iy = iyy + ky - 1;
    const bool rowok = iy >= 0 && iy < p.H;
    const T* xrow = x + (((long long)ib * p.H + (rowok ? iy : iyy)) * p.W + ix * KP) * p.Cin;
    const T* drow = dy + (((long long)ib * p.H + iyy) * p.W + ix * KP) * p.Cout;
    const int first = ix == 0 ? 1 : 0, last = ix == tiles_x - 1 ? 1 : 0;
    char* base = smem + soff;
#pragma unroll
    for (int i = 0; i < NCW; ++i) {
      const T* src = (d_isx[i] ? xrow : drow) + (d_off[i] + (i == 0 ? first * d_dl : 0) + (i == I_R ? last * d_dr : 0));
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

  // ---- NRM: the operand's pending norm + ReLU, applied IN LDS to the landed input tile of the NEXT step while this step's MFMAs run
  // (conv_wgrad3_kernel does it in registers between the global load and the ds_write: same arithmetic, bit-identical tile).  Thread
  // t owns the 8-byte half slots t, t + 256 and (t < 32) t + 512 of the 34 x 16 half slots: the same four channels in all three.
  const int nr_row = tid >> 4;                                                       // pixel row of the first half slot (then + 16, + 32)
  const int nr_ch = ci0 + 8 * (((tid & 15) >> 1) ^ (((nr_row >> 1) & 1) << 2)) + 4 * (tid & 1);      // (the swizzle repeats every 4 rows)
  float mu[4], sc[4], be[4];
  int state_b = -1;
  int nb = 0, ny = 0, nx = 0;                // tile coordinates of the step whose input tile is normalised next
  if constexpr (NRM) { nb = cb; ny = cy; nx = cx; }
  // The norm states of the samples this split touches are copied to LDS BEFORE the first DMA is issued and re-read from there on a
  // change of sample: a global load inside the loop would make the compiler's wait-count pass put vmcnt(0) -- i.e. a wait for every
  // prefetch in flight -- in front of the norm arithmetic of every step (measured: 100 us instead of 78).
  const int nr_b0 = cb;
  if constexpr (NRM) {
    const size_t plane = (size_t)p.B * p.Cin;
    float* stl = reinterpret_cast<float*>(smem + NS * STAGE);
    const int lastb = (step0 + (nsteps > 0 ? nsteps - 1 : 0)) / (p.H * tiles_x);
    const int cnt = (lastb - nr_b0 + 1) * 3 * BM;
    for (int idx = tid; idx < cnt; idx += 256) {
      const int sb = idx / (3 * BM), k = (idx / BM) % 3, ch = idx % BM;
      stl[idx] = p.in_norm[(size_t)(k == 0 ? 0 : k + 1) * plane + (size_t)(nr_b0 + sb) * p.Cin + ci0 + ch];
    }
    __syncthreads();
  }
  auto nr_state = [&]() {                   // (a change of sample: at most every H * W / 32 steps)
    if (nb != state_b) {
      typedef __attribute__((address_space(3))) char* lds_ptr;
      const unsigned a = (unsigned)(size_t)(lds_ptr)(smem + NS * STAGE) + (nb - nr_b0) * (3 * BM * 4) + (nr_ch - ci0) * 4;
      f32x4 q[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) asm volatile("ds_read_b128 %0, %1" : "=v"(q[k]) : "v"(a + k * BM * 4) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2])::"memory");
#pragma unroll
      for (int e = 0; e < 4; ++e) { mu[e] = q[0][e]; sc[e] = q[1][e]; be[e] = q[2][e]; }
      state_b = nb;
    }
  };
  auto nr_math = [&](f32x2 raw, bool zero) {
    bf16x4_w v = __builtin_bit_cast(bf16x4_w, raw), o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float f = fmaf((float)v[e] - mu[e], sc[e], be[e]);
      if (p.relu_in) f = fmaxf(f, 0.f);
      o[e] = zero ? (bf16_t)0.f : (bf16_t)f;
    }
    return __builtin_bit_cast(f32x2, o);
  };
  const bool nr_three = tid < (XROWS - 32) * 16;       // rows 32 and 33

  typedef __attribute__((address_space(3))) char* lds_ptr;
  const unsigned lds0 = (unsigned)(size_t)(lds_ptr)smem;
  static_assert(MR == 1, "one 32-channel input fragment row per wave");
  const unsigned nrl = lds0 + tid * 8;          // NRM: the thread's first half slot of a stage's input tile
  unsigned xl[3], yl[NR];                      // lane parts of the fragment addresses (input tile per tap, dy tile per 32-channel block)
#pragma unroll
  for (int t = 0; t < 3; ++t) xl[t] = lds0 + wfrag_lane<XB>(t, wm * 32, lane);
#pragma unroll
  for (int j = 0; j < NR; ++j) yl[j] = lds0 + wfrag_lane<YB>(0, wn * NR * 32 + j * 32, lane);

  // one K step on the ring stage at byte offset soff (noff: the next step's stage, ioff: the stage the step issues into).  ONE copy of
  // the loop body, stage offsets in scalar registers: the stage-unrolled form (every address a compile-time constant) kept one set of
  // address registers per stage alive and spilled at five stages
  auto body = [&](int st, int soff, int noff, int ioff) {
    // this wave's pieces of step st (NRM: st + 1, whose input tile is normalised during this step) have landed -- those of the
    // steps issued after it may still be in flight -- and every LDS access it issued has retired: a read that is merely issued can
    // still be queued when another wave's DMA lands on the bytes
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    {
      // issued so far: the steps up to st + D - 1; needed now: step st (NRM: st + 1); whatever was issued after that may stay in flight
      const int last = st + D - 1 < nsteps - 1 ? st + D - 1 : nsteps - 1;
      const int ahead = last - (NRM ? st + 1 : st);
      constexpr int W2 = 2 * NCW, W1 = NCW;
      if (ahead >= 2) __builtin_amdgcn_s_waitcnt((W2 & 15) | ((W2 >> 4) << 14) | 0x0070);
      else if (ahead == 1) __builtin_amdgcn_s_waitcnt((W1 & 15) | ((W1 >> 4) << 14) | 0x0070);
      else __builtin_amdgcn_s_waitcnt(0x0070);
    }
    const int iy = cy + ky - 1;
    const bool rowok = iy >= 0 && iy < p.H;
    if constexpr (!NRM) {
      if (wid == 0 && lane < XB / 16) {            // wave 0 issued the pieces that hold input pixels 0 and 33 (inline asm: see wfrag)
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const unsigned a0 = lds0 + lane * 16 + soff;
        if (cx == 0) asm volatile("ds_write_b128 %0, %1" ::"v"(a0), "v"(z) : "memory");
        if (cx == tiles_x - 1) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(a0), "v"(z), "n"((XROWS - 1) * XB) : "memory");
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
#ifdef PWR_DEBUG_BUILD
    if (st + D < nsteps && !(p.dbg & 4)) issue(ioff);
    const bool domma = rowok && !(p.dbg & 2);
#else
    if (st + D < nsteps) issue(ioff);                       // into the stage that was read during step st - 1
    const bool domma = rowok;
#endif
    V bf[2][NR], af[2][3][MR];
    constexpr int NF = 2 * (NR + 3 * MR);          // LDS reads of one half step
    static_assert(NF + 1 <= 15, "lgkmcnt range");
#define PWR_FRAGS(ss)                                                                                                     \
  _Pragma("unroll") for (int j = 0; j < NR; ++j) bf[ss][j] = wfrag<YB, XBYTES + ss * 16 * YB>(yl[j] + soff);                \
  _Pragma("unroll") for (int t = 0; t < 3; ++t) af[ss][t][0] = wfrag<XB, ss * 16 * XB>(xl[t] + soff);
    // the half step's fragments have arrived when at most CNT later LDS accesses are outstanding; tied to the fragment registers so
    // that no MFMA can be scheduled above the wait
#define PWR_ARRIVED(ss, CNT)                                                                                              \
  _Pragma("unroll") for (int j = 0; j < NR; ++j) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(bf[ss][j]) : "n"(CNT) : "memory"); \
  _Pragma("unroll") for (int t = 0; t < 3; ++t) _Pragma("unroll") for (int i = 0; i < MR; ++i) asm volatile("" : "+v"(af[ss][t][i]));
#define PWR_MMAS(ss)                                                                                                      \
  _Pragma("unroll") for (int t = 0; t < 3; ++t) _Pragma("unroll") for (int i = 0; i < MR; ++i) _Pragma("unroll") for (int j = 0; j < NR; ++j) \
      acc[t][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ss][t][i], bf[ss][j], acc[t][i][j], 0, 0, 0);
#ifdef PWR_DEBUG_BUILD
    const bool mm = domma && !(p.dbg & 1);
#else
    const bool mm = domma;
#endif
    if constexpr (NRM) {
      // The raw input tile of step st + 1 (ring stage + 1) is read back, normalised and stored in place; the round trip hides
      // behind this step's first fragment reads, the stores are visible after the next barrier.  STRAIGHT-LINE on purpose -- the tile
      // of an out-of-image row (nobody consumes it) and, at the last step, a stage that holds no tile are transformed all the same:
      // the results of the inline-asm reads arrive asynchronously, and a register copy the compiler places in front of the wait (it
      // did, where two paths with their own waits merged) copies stale values.
      if (st + 1 < nsteps) nr_state();
      constexpr int NOFF = 0;
      const unsigned nra = nrl + noff;
      f32x2 r0, r1, r2;                      // (threads 32 .. 255 read the third value from beyond the tile and do not store it)
      asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r0) : "v"(nra), "n"(NOFF) : "memory");
      asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r1) : "v"(nra), "n"(NOFF + 2048) : "memory");
      asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r2) : "v"(nra), "n"(NOFF + 4096) : "memory");
      PWR_FRAGS(0)
      PWR_FRAGS(1)
      // the raw tile values and the first half's fragments have arrived when only the second half's reads are outstanding; the norm
      // arithmetic then issues in the shadow of the first half's MFMAs
      asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(r0), "+v"(r1), "+v"(r2) : "n"(NF) : "memory");
      PWR_ARRIVED(0, NF)
      auto nr_store = [&]() {
#ifdef PWR_DEBUG_BUILD
        if (p.dbg & 128) return;                       // elimination: no norm arithmetic, no stores
        if (p.dbg & 64) {                              // elimination: stores of the raw values, no arithmetic
          asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(nra), "v"(r0), "n"(NOFF) : "memory");
          asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(nra), "v"(r1), "n"(NOFF + 2048) : "memory");
          return;
        }
#endif
        const f32x2 o0 = nr_math(r0, nr_row == 0 && nx == 0);
        asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(nra), "v"(o0), "n"(NOFF) : "memory");
        const f32x2 o1 = nr_math(r1, false);
        asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(nra), "v"(o1), "n"(NOFF + 2048) : "memory");
        if (nr_three) {
          const f32x2 o2 = nr_math(r2, nx == tiles_x - 1 && nr_row == XROWS - 33);
          asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(nra), "v"(o2), "n"(NOFF + 4096) : "memory");
        }
      };
      if (mm) {                                // (same basic block as the MFMAs, so that the scheduler can interleave the two)
        PWR_MMAS(0)
        nr_store();
#pragma unroll
        for (int g = 0; g < 3 * MR * NR; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 48 / (3 * MR * NR), 0);
        }
      } else {
        nr_store();
      }
      if (st + 1 < nsteps) { if (++nx == tiles_x) { nx = 0; if (++ny == p.H) { ny = 0; ++nb; } } }
      PWR_ARRIVED(1, 2)
      if (mm) { PWR_MMAS(1) }
    } else if (domma) {
      PWR_FRAGS(0)
      PWR_FRAGS(1)
      PWR_ARRIVED(0, NF)
      if (mm) { PWR_MMAS(0) }
      __builtin_amdgcn_sched_barrier(0);
      PWR_ARRIVED(1, 0)
      if (mm) { PWR_MMAS(1) }
    }
#undef PWR_FRAGS
#undef PWR_ARRIVED
#undef PWR_MMAS
    if (++cx == tiles_x) { cx = 0; if (++cy == p.H) { cy = 0; ++cb; } }
  };

  if (nsteps > 0) {
#pragma unroll
    for (int k = 0; k < D; ++k)
      if (k < nsteps) issue(k * STAGE);
    if constexpr (NRM) {                       // the tile of step 0: landed everywhere, normalised, before the loop's first barrier
      constexpr int W3 = 3 * NCW, W2 = 2 * NCW, W1 = NCW;
      static_assert(D == 4, "prologue waits");
      if (nsteps > 3) __builtin_amdgcn_s_waitcnt((W3 & 15) | ((W3 >> 4) << 14) | 0x0070);
      else if (nsteps > 2) __builtin_amdgcn_s_waitcnt((W2 & 15) | ((W2 >> 4) << 14) | 0x0070);
      else if (nsteps > 1) __builtin_amdgcn_s_waitcnt((W1 & 15) | ((W1 >> 4) << 14) | 0x0070);
      else __builtin_amdgcn_s_waitcnt(0x0070);
      __builtin_amdgcn_s_barrier();
      __atomic_signal_fence(__ATOMIC_SEQ_CST);
      nr_state();
      {
        const unsigned na0 = nrl;
        f32x2 r0, r1, r2;
        asm volatile("ds_read_b64 %0, %1" : "=v"(r0) : "v"(na0) : "memory");
        asm volatile("ds_read_b64 %0, %1 offset:2048" : "=v"(r1) : "v"(na0) : "memory");
        asm volatile("ds_read_b64 %0, %1 offset:4096" : "=v"(r2) : "v"(na0) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2)::"memory");
        const f32x2 o0 = nr_math(r0, nr_row == 0 && nx == 0);
        asm volatile("ds_write_b64 %0, %1" ::"v"(na0), "v"(o0) : "memory");
        const f32x2 o1 = nr_math(r1, false);
        asm volatile("ds_write_b64 %0, %1 offset:2048" ::"v"(na0), "v"(o1) : "memory");
        if (nr_three) {
          const f32x2 o2 = nr_math(r2, nx == tiles_x - 1 && nr_row == XROWS - 33);
          asm volatile("ds_write_b64 %0, %1 offset:4096" ::"v"(na0), "v"(o2) : "memory");
        }
      }
      if (++nx == tiles_x) { nx = 0; if (++ny == p.H) { ny = 0; ++nb; } }
    }
    int stg = 0;
    for (int st = 0; st < nsteps; ++st) {
      const int nstg = stg + 1 == NS ? 0 : stg + 1, istg = stg == 0 ? NS - 1 : stg - 1;      // (stg + D) % NS
      body(st, stg * STAGE, nstg * STAGE, istg * STAGE);
      stg = nstg;
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

// 3x3 stride 1, 32-pixel row segments, 64-channel tiles.  PWR_WGRAD3_DMA (debug build): 0 never, 1 only operands without a pending
// norm, 2 also the operands whose norm + ReLU is applied in LDS, 3 (shipped) those only for 64-wide output tiles (the 128-wide
// tile is faster register-staged: 75.9 vs 85.8 us at the C2 heads shape; 64 -> 64 at 128 x 128, B = 8: 108 vs 85 us)
bool wgrad3d_applicable(const WgradParams& p) {
  static const int on = PWR_DBG_ENV("PWR_WGRAD3_DMA", 3);
  if (p.in_norm && (on < 2 || (on == 3 && p.Cout % 128 == 0) || p.steps_per_split > 3 * (p.H * p.W / 32))) return false;      // (a split spans at most 4 samples' norm states)
  return on != 0 && p.ksize == 3 && p.stride == 1 && p.W % 32 == 0 && p.M % 32 == 0 && p.Cin % 64 == 0 &&
         p.Cout % 64 == 0 && p.CoutPad == p.Cout;
}

int launch_wgrad3d(const WgradParams& p0, hipStream_t s) {
  WgradParams p = p0;
  p.dbg = PWR_DBG_ENV("PWR_WGRAD3D_DBG", 0);
  const int bn = p.Cout % 128 == 0 ? 128 : 64;
  dim3 grid(24 * ((p.S + 7) / 8), (p.Cin / 64) * (p.CoutPad / bn), 1), block(256);
  if (p.in_norm) {
    if (bn == 128) hipLaunchKernelGGL((conv_wgrad3d_kernel<64, 128, true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((conv_wgrad3d_kernel<64, 64, true>), grid, block, 0, s, p);
  } else {
    if (bn == 128) hipLaunchKernelGGL((conv_wgrad3d_kernel<64, 128, false>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((conv_wgrad3d_kernel<64, 64, false>), grid, block, 0, s, p);
  }
  return (int)hipGetLastError();
}

}  // namespace pwr
