// bf16 weight gradient of a 3x3 stride-1 conv, wave-specialised, NINE taps per workgroup (round 4, second form).
//
//   dW[co][ci][ky][kx] = sum_{b, y, x} a[b][y + ky - 1][x + kx - 1][ci] * dy[b][y][x][co]          (/root/reference/model.py:55-63 /
//   :104-112: the weight gradients of the heads' 128 -> 128 convs under autograd; a = the conv's input AFTER its norm + ReLU)
//
// conv_wgrad_ws.hip (three taps of one kernel row per workgroup, 128 x 128 channels) turned out to be bound by what ONE CU can pull in by
// LDS-DMA -- 25 - 30 GB/s in this access pattern (tools/wgrad_ws_stamps.py: its MFMA waves spend 45 - 55 % of every K step waiting in
// the barrier for the loader waves, which spend theirs issuing 17 KiB of DMA; profiles/r4_experiments.md section 7) -- and, with a norm
// on the operand, by normalising every input row three times (once per kernel-row workgroup).  This form restructures the work so that
// a K step needs HALF the bytes per FLOP and a THIRD of the norm arithmetic:
//   workgroup  = (split, 64 ci x 64 co tile), all NINE taps: 4 MFMA waves x (32 ci x 32 co x 9 taps) = 144 accumulator registers
//   K walk     = DOWN the image columns: the K step of output row y needs the input rows y - 1, y, y + 1 of its 32-pixel segment; two of
//                them are already in LDS from the steps of rows y - 1 and y - 2, so a step brings in ONE new input row (34 px x 64 ch =
//                4.3 KiB, normalised once, used by three steps) and one dy tile (32 px x 64 ch = 4 KiB): 8.3 KiB per 2.36 MFLOP
//                against 17 KiB per 3.1 MFLOP.
// The K stream is a sequence of EVENTS, H + 2 per image column (batch x 32-pixel segments): event k of a column brings in input row
// k - 1 (rows -1 and H are the conv's zero padding: a zero-filled slot) and, for k >= 2, multiplies dy row k - 2 with the three newest
// input rows.  Events 0 and 1 of a column (and the two lead-in events of a split that starts in mid-column) only load: their dy tile is
// zero-filled, so the MFMA waves run ONE branch-free loop body (18 MFMAs on zeros, 3 % of the events).  Everything else as in
// conv_wgrad_ws.hip: 4 loader waves (LDS-DMA into XOR-swizzled plain rows, the norm + ReLU pass in LDS one event before a row is read,
// counted vmcnt, one barrier per event), MFMA waves that issue no vector-memory instruction and read LDS through compiler-visible
// transposing reads, fp32 split-K slabs in the layout of the common reduce.  The order of summation over K differs from the
// row-major kernels', so results are not bit-identical to theirs (they are deterministic, and equal between the norm and no-norm form).
//
// MEASURED SLOWER than conv_wgrad_ws.hip (profiles/r4_experiments.md section 7: 2.36 MFLOP per barrier interval is too little -- the
// interval has a floor of ~0.3 us of barrier + loop control that the three-tap kernel spreads over 3.1 MFLOP -- isolated 67 against 56 us
// without a norm, in the train step 5.75 against 5.70 ms at its best workgroup count).  It is an EXPERIMENT: compiled into the DEBUG build
// only (tools/build_debug.py picks up tools/csrc_debug/*.hip, PWR_WGRAD9W=1 selects it); the shipped library does not contain it.
#include <cstdlib>

#include "conv_common.h"
#include "pwr.h"

#ifndef PWR_DEBUG_BUILD
#error "debug-build source (tools/build_debug.py): not part of the product library"
#endif

namespace pwr {

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_w9;

namespace w9 {
constexpr int KP = 32, XPIX = KP + 2, RB = 128;                  // K step (pixels), staged input pixels per row, bytes per staged pixel (64 ch bf16)
constexpr int XCH = 5, YCH = 4, NCH = XCH + YCH;                 // 1-KiB DMA pieces per event: one input row (34 x 128 B = 4.25 KiB), one dy tile (4 KiB)
constexpr int NLW = 4;                                           // loader waves: wave 0 issues pieces 0, 4, 8; the others lw, lw + 4
constexpr int XSLOT = XCH * 1024, YSTAGE = YCH * 1024;
constexpr int D = 5;                                             // an event's pieces are issued D events ahead
constexpr int NRX = D + 3, NSD = D + 1;                          // input-row slots (a row is read by three events), dy stages
constexpr int XREGION = NRX * XSLOT;
constexpr int MAXSB = 8;                                         // norm states of at most this many samples per split
constexpr int STATE_OFF = XREGION + NSD * YSTAGE;
constexpr int LDS_BYTES = STATE_OFF + MAXSB * 3 * 64 * 4;

// byte offset of 16-byte slot `slot` of pixel row `row` (128-byte rows: conv_wgrad_dma.hip's wswz<128>)
__device__ __forceinline__ int swz(int row, int slot) { return row * RB + ((slot ^ (((row >> 1) & 1) << 2)) << 4); }
// lane part of the address of an MFMA 32x32x16 operand fragment read by two ds_read_b64_tr_b16 (rows +0 / +4)
__device__ __forceinline__ int frag_lane(int k0, int chb, int lane) {
  const int li = lane & 15, cg = (lane >> 4) & 1, h = lane >> 5, q = li >> 2, pp = li & 3;
  return swz(k0 + 8 * h + q, (chb >> 3) + 2 * cg + (pp >> 1)) + 8 * (pp & 1);
}
__device__ __forceinline__ bf16x8 frag(const char* a) {
  typedef __attribute__((address_space(3))) bf16x4_w9* lptr;
  const bf16x4_w9 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lptr)a);
  const bf16x4_w9 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lptr)(a + 4 * RB));
  bf16x8 f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3]; f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  return f;
}
constexpr unsigned vmwait(int n) { return (unsigned)((n & 15) | ((n >> 4) << 14) | 0x0070); }      // s_waitcnt vmcnt(n) lgkmcnt(0)
}  // namespace w9

template <bool NRM, bool RELU>
__global__ __launch_bounds__(512, 2) void conv_wgrad9w_kernel(WgradParams p) {
  using namespace w9;
  typedef bf16_t T;
  typedef bf16x8 V;
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  // blockIdx.x enumerates (split, tile) so that all tiles of a split are 8 ids apart (one XCD, speed only: they share the split's input
  // rows and dy tiles in that XCD's L2)
  const int ntn = p.Cout / 64, ntiles = (p.Cin / 64) * ntn;
  const int q_ = blockIdx.x >> 3;
  const int tile = q_ % ntiles;
  const int split = (q_ / ntiles) * 8 + (blockIdx.x & 7);
  if (split >= p.S) return;
  const int ci0 = (tile / ntn) * 64, co0 = (tile % ntn) * 64;
  const int tiles_x = p.W / KP, EV = p.H + 2;                    // events per image column
  const int total = p.B * tiles_x * EV;
  const int eps = (total + p.S - 1) / p.S;
  const int c0 = split * eps < total ? split * eps : total;       // events [c0, c1) are this split's to COMPUTE
  const int c1 = c0 + eps < total ? c0 + eps : total;
  const int g0 = c0 >= 2 ? c0 - 2 : 0;                            // ... after loading from two events earlier (the input rows they need)
  const int nev = c1 - g0 > 0 && c1 > c0 ? c1 - g0 : 0;
  // coordinates of the stream's first event
  const int col0 = g0 / EV, k0 = g0 - col0 * EV;
  const int b0 = col0 / tiles_x, x0 = col0 - b0 * tiles_x;

  if constexpr (NRM) {
    // the norm states (mean, scale, beta) of the samples this split touches: to LDS BEFORE the first DMA is issued (conv_wgrad_ws.hip)
    const size_t plane = (size_t)p.B * p.Cin;
    float* stl = reinterpret_cast<float*>(smem + STATE_OFF);
    const int lastb = nev > 0 ? (c1 - 1) / (tiles_x * EV) : b0;
    const int cnt = (lastb - b0 + 1) * 3 * 64;
    for (int idx = tid; idx < cnt; idx += 512) {
      const int sb = idx / (3 * 64), k = (idx >> 6) % 3, ch = idx & 63;
      stl[idx] = p.in_norm[(size_t)(k == 0 ? 0 : k + 1) * plane + (size_t)(b0 + sb) * p.Cin + ci0 + ch];
    }
    __syncthreads();
  }

  if (wid >= NLW) {
    // =================================================================== loader waves
    const int lw = wid - NLW, lt = tid - 64 * NLW;
    const T* __restrict__ x = reinterpret_cast<const T*>(p.x);
    const T* __restrict__ dy = reinterpret_cast<const T*>(p.dy);
    // per-lane descriptors of this wave's DMA pieces: pieces lw, lw + 4 (and 8 for wave 0).  Pieces 0 .. 4 = the input row (pixel 0 in
    // piece 0, pixel 33 and the unused tail in piece 4: both wave 0's), 5 .. 8 = the dy tile
    constexpr int NPW = 3;
    int d_lds[NPW], d_off[NPW], d_dl = 0, d_dr = 0;
    bool d_isx[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      int c = lw + NLW * i;
      if (c >= NCH) c = lw;                         // (not issued: only wave 0 has a third piece)
      d_isx[i] = c < XCH;
      const int cx_ = c < XCH ? c : c - XCH;
      d_lds[i] = d_isx[i] ? cx_ * 1024 : cx_ * 1024;
      const int q = 64 * cx_ + lane;
      const int r0 = q >> 3, s1 = q & 7;              // pixel row and physical slot of this lane's 16 bytes
      const int xr = r0 < XPIX ? r0 : XPIX - 1;       // (beyond the 34th pixel: nobody reads it)
      const int sl = s1 ^ (((r0 >> 1) & 1) << 2);     // the channel slot that belongs there
      d_off[i] = d_isx[i] ? (xr - 1) * p.Cin + ci0 + 8 * sl : r0 * p.Cout + co0 + 8 * sl;
      if (i == 0) d_dl = (d_isx[i] && xr == 0) ? p.Cin : 0;              // out-of-image halo pixel: clamped into the row, zeroed in LDS afterwards
      if (i == 1) d_dr = (d_isx[i] && xr == XPIX - 1) ? -p.Cin : 0;
    }
    const bool three = lw == 0;                         // (wave-uniform)
    int ib = b0, ixs = x0, ik = k0, issued = 0, ixslot = 0, iystage = 0;      // next event to ISSUE, and where it goes
    auto issue = [&]() {
      // input row ik - 1 (out of the image: a clamped row, zero-filled by the pass) and dy row ik - 2 (events 0, 1: a clamped row, zero-filled)
      const int r = ik - 1, yy = ik - 2;
      const int rc = r < 0 ? 0 : (r >= p.H ? p.H - 1 : r), yc = yy < 0 ? 0 : yy;
      const T* xrow = x + (((long long)ib * p.H + rc) * p.W + ixs * KP) * p.Cin;
      const T* drow = dy + (((long long)ib * p.H + yc) * p.W + ixs * KP) * p.Cout;
      const int first = ixs == 0 ? 1 : 0, last = ixs == tiles_x - 1 ? 1 : 0;
      char* xb = smem + ixslot * XSLOT;
      char* yb = smem + XREGION + iystage * YSTAGE;
#pragma unroll
      for (int i = 0; i < NPW; ++i) {
        if (i == NPW - 1 && !three) break;
        const T* src = (d_isx[i] ? xrow : drow) + (d_off[i] + (i == 0 ? first * d_dl : 0) + (i == 1 ? last * d_dr : 0));
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)((d_isx[i] ? xb : yb) + d_lds[i]), 16, 0, 0);
      }
      ++issued;
      ixslot = ixslot + 1 == NRX ? 0 : ixslot + 1;
      iystage = iystage + 1 == NSD ? 0 : iystage + 1;
      if (++ik == EV) { ik = 0; if (++ixs == tiles_x) { ixs = 0; ++ib; } }
    };
    // all but this wave's pieces of the `k` most recently issued events have landed, and every LDS access of the wave has retired
    auto landed_but = [&](int k) {
      __atomic_signal_fence(__ATOMIC_SEQ_CST);
      static_assert(D - 2 <= 3, "wait cases below");
      if (three) {
        if (k >= 3) __builtin_amdgcn_s_waitcnt(vmwait(9));
        else if (k == 2) __builtin_amdgcn_s_waitcnt(vmwait(6));
        else if (k == 1) __builtin_amdgcn_s_waitcnt(vmwait(3));
        else __builtin_amdgcn_s_waitcnt(vmwait(0));
      } else {
        if (k >= 3) __builtin_amdgcn_s_waitcnt(vmwait(6));
        else if (k == 2) __builtin_amdgcn_s_waitcnt(vmwait(4));
        else if (k == 1) __builtin_amdgcn_s_waitcnt(vmwait(2));
        else __builtin_amdgcn_s_waitcnt(vmwait(0));
      }
      __atomic_signal_fence(__ATOMIC_SEQ_CST);
    };

    // ---- the in-LDS pass over a landed event: norm + ReLU of the input row (NRM), zeros for its out-of-image halo pixels, an all-zero
    // row where the whole row is the conv's zero padding, and an all-zero dy tile for the events that only load.  Loader thread lt owns
    // the 16-byte slot lt of the row's 34 x 8 slots (pixel lt / 8); the 16 slots of pixels 32 and 33 go to the threads 192 .. 207
    // (pixels 24, 25 + 8: the SAME eight channels, the swizzle repeats every 4 pixels)
    typedef __attribute__((address_space(3))) char* lds_ptr;
    const unsigned lds0 = (unsigned)(size_t)(lds_ptr)smem;
    const unsigned nrl = lds0 + lt * 16;
    const int nr_px = lt >> 3;
    const int nr_ch = 8 * ((lt & 7) ^ (((nr_px >> 1) & 1) << 2));      // first of the eight channels (relative to ci0)
    const bool second = lt >= 192 && lt < 208;                        // (wave 3 only)
    constexpr int OFF2 = (256 - 192) * 16;
    float mu[8], sc[8], be[8];
    int state_b = -1;
    int nb = b0, nxs = x0, nk = k0, nev_g = g0, nxslot = 0, nystage = 0;      // the event that is passed next
    auto nr_state = [&]() {
      if (nb != state_b) {
        const unsigned a = lds0 + STATE_OFF + (nb - b0) * (3 * 64 * 4) + nr_ch * 4;
        f32x4 q0, q1, q2, q3, q4, q5;
        asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:16\n\tds_read_b128 %2, %6 offset:256\n\tds_read_b128 %3, %6 offset:272\n\t"
                     "ds_read_b128 %4, %6 offset:512\n\tds_read_b128 %5, %6 offset:528\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&v"(q4), "=&v"(q5) : "v"(a) : "memory");
#pragma unroll
        for (int e = 0; e < 4; ++e) { mu[e] = q0[e]; mu[4 + e] = q1[e]; sc[e] = q2[e]; sc[4 + e] = q3[e]; be[e] = q4[e]; be[4 + e] = q5[e]; }
        state_b = nb;
      }
    };
    auto nr_math = [&](f32x4 raw) {                               // conv_wgrad3_kernel's arithmetic
      V v = __builtin_bit_cast(V, raw), o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float f = fmaf((float)v[e] - mu[e], sc[e], be[e]);
        if constexpr (RELU) f = fmaxf(f, 0.f);
        o[e] = (bf16_t)f;
      }
      return __builtin_bit_cast(f32x4, o);
    };
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto event_pass = [&]() {
      const bool zl = nxs == 0, zr = nxs == tiles_x - 1;          // the column touches the left / right image border
      const int r = nk - 1;
      const bool rowzero = r < 0 || r >= p.H;                     // the input row is zero padding
      const bool nodot = nk < 2 || nev_g < c0;                    // the event only loads: its dy tile must read as zeros
      const unsigned a = nrl + nxslot * XSLOT;
      if constexpr (NRM) {
        nr_state();
        f32x4 r0, r1;
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:%3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r0), "=&v"(r1) : "v"(a), "n"(OFF2) : "memory");
        f32x4 o0 = nr_math(r0);
        if (rowzero | zl) {                                        // (wave-uniform, rare: image borders)
          if (rowzero || nr_px == 0) o0 = zero4;
        }
        asm volatile("ds_write_b128 %0, %1" ::"v"(a), "v"(o0) : "memory");
        if (lw == NLW - 1) {                                       // (wave-uniform: the wave that owns pixels 32 and 33)
          f32x4 o1 = nr_math(r1);
          if (rowzero || (zr && lt >= 200)) o1 = zero4;            // (threads 200 .. 207 hold pixel 33)
          if (second) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(a), "v"(o1), "n"(OFF2) : "memory");
        }
      } else {
        if (rowzero) {
          asm volatile("ds_write_b128 %0, %1" ::"v"(a), "v"(zero4) : "memory");
          if (second) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(a), "v"(zero4), "n"(OFF2) : "memory");
        } else if (lt < 8) {                                       // one 128-byte pixel = 8 slots
          if (zl) asm volatile("ds_write_b128 %0, %1" ::"v"(a), "v"(zero4) : "memory");
          if (zr) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(a), "v"(zero4), "n"((XPIX - 1) * RB) : "memory");
        }
      }
      if (nodot) {                                                 // (wave-uniform: 2 of H + 2 events per column, 2 per split)
        const unsigned ay = nrl + XREGION + nystage * YSTAGE;      // 256 slots of 16 bytes: one per loader thread
        asm volatile("ds_write_b128 %0, %1" ::"v"(ay), "v"(zero4) : "memory");
      }
      ++nev_g;
      nxslot = nxslot + 1 == NRX ? 0 : nxslot + 1;
      nystage = nystage + 1 == NSD ? 0 : nystage + 1;
      if (++nk == EV) { nk = 0; if (++nxs == tiles_x) { nxs = 0; ++nb; } }
    };

    // ---- prologue: events 0 .. D-1 in flight, events 0 and 1 passed, event 2 landed
#pragma unroll
    for (int k = 0; k < D; ++k)
      if (k < nev) issue();
    // the row slots of the two events BEFORE the stream's first one are read by the events 0 and 1 (against their zero dy tile): they must
    // hold finite values, not whatever the LDS held (0 x NaN = NaN)
    for (int q = lt; q < 2 * XSLOT / 16; q += 64 * NLW) asm volatile("ds_write_b128 %0, %1" ::"v"(lds0 + (NRX - 2) * XSLOT + q * 16), "v"(zero4) : "memory");
    landed_but(issued - 2);
    __builtin_amdgcn_s_barrier();                                  // (P) every loader's pieces of events 0 and 1 have landed
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    if (nev > 0) event_pass();
    if (nev > 1) event_pass();
    landed_but(issued - 3);
#ifdef PWR_DEBUG_BUILD
    long long tb = 0, tv = 0, t_loop0 = (long long)__builtin_amdgcn_s_memtime();
#endif
#pragma nounroll
    for (int s = 0; s < nev; ++s) {
#ifdef PWR_DEBUG_BUILD
      const long long tb0 = p.stamps ? (long long)__builtin_amdgcn_s_memtime() : 0;
#endif
      __builtin_amdgcn_s_barrier();                                // barrier s: events <= s + 1 passed, event s + 2 landed everywhere
#ifdef PWR_DEBUG_BUILD
      if (p.stamps) tb += (long long)__builtin_amdgcn_s_memtime() - tb0;
#endif
      __atomic_signal_fence(__ATOMIC_SEQ_CST);
#ifdef PWR_DEBUG_BUILD
      if (s + D < nev && !(p.dbg & 2)) issue();                     // (elimination: no DMA after the prologue)
#else
      if (s + D < nev) issue();                                    // into the row slot last read during event s - 1 and the dy stage of event s - 1
#endif
      if (s + 2 < nev) event_pass();
#ifdef PWR_DEBUG_BUILD
      const long long tv0 = p.stamps ? (long long)__builtin_amdgcn_s_memtime() : 0;
#endif
      // issued so far: the events up to min(s + D, nev - 1); needed at barrier s + 1: event s + 3 landed (and every LDS store retired)
      landed_but((s + D < nev - 1 ? s + D : nev - 1) - (s + 3));
#ifdef PWR_DEBUG_BUILD
      if (p.stamps) tv += (long long)__builtin_amdgcn_s_memtime() - tv0;
#endif
    }
#ifdef PWR_DEBUG_BUILD
    if (p.stamps && lane == 0) {
      long long* d = p.stamps + (size_t)blockIdx.x * 32 + wid * 4;
      d[0] = tb; d[1] = tv; d[2] = (long long)__builtin_amdgcn_s_memtime() - t_loop0; d[3] = nev;
    }
#endif
    __builtin_amdgcn_s_waitcnt(vmwait(0));                         // (no DMA may be in flight when the workgroup's LDS is released)
    __builtin_amdgcn_s_barrier();                                  // barrier nev (the MFMA waves' last)
    return;
  }

  // ===================================================================== MFMA waves
  const int wm = wid >> 1, wn = wid & 1;
  f32x16 acc[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
  // lane parts of the fragment addresses: input row per kx (32-channel block wm), dy tile (32-channel block wn)
  const char* xl[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) xl[kx] = smem + frag_lane(kx, wm * 32, lane);
  const char* yl = smem + XREGION + frag_lane(0, wn * 32, lane);

  // unit u = (K half h = u / 3, kernel row ky = u % 3): three input fragments (kx = 0, 1, 2 of the row of event s - 2 + ky), three MFMAs
  // against the dy fragment of half h.  Input fragments in a ring of three units (read two units ahead), dy fragments double-buffered.
  V A[3][3], Bf[2];
  auto loadA = [&](V (&a)[3], int slotoff, int h) {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) a[kx] = frag(xl[kx] + slotoff + h * 16 * RB);
  };
  auto loadB = [&](V& b, int yoff, int h) { b = frag(yl + yoff + h * 16 * RB); };
  auto mma = [&](int ky, const V (&a)[3], const V& b) {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) acc[ky][kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kx], b, acc[ky][kx], 0, 0, 0);
  };
#define PWR_W9_SCHED(reads)                                  \
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       \
  __builtin_amdgcn_sched_group_barrier(0x100, reads, 0);   \
  __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);

  __syncthreads();                                                 // (P)
  __syncthreads();                                                 // barrier 0: events 0 and 1 are passed
  // row-slot offsets of the events s - 2, s - 1, s (the rows ky = 0, 1, 2 of event s) and s + 1; events before the stream's first one do
  // not exist: their "rows" are whatever the slots NRX - 2, NRX - 1 hold -- multiplied with a zero dy tile (events 0, 1 never compute)
  int o0 = (NRX - 2) * XSLOT, o1 = (NRX - 1) * XSLOT, o2 = 0, o3 = XSLOT, ys = 0, ysn = YSTAGE;
  loadA(A[0], o0, 0); loadA(A[1], o1, 0); loadB(Bf[0], ys, 0);
#ifdef PWR_DEBUG_BUILD
  long long tbm = 0, t_loopm = (long long)__builtin_amdgcn_s_memtime();
#endif
#pragma nounroll
  for (int s = 0; s < nev; ++s) {
#ifdef PWR_DEBUG_BUILD
    if (p.dbg & 1) { __syncthreads(); continue; }                  // (elimination: the MFMA waves only keep the barriers)
#endif
    loadA(A[2], o2, 0); loadB(Bf[1], ys, 1); mma(0, A[0], Bf[0]); PWR_W9_SCHED(8)
    loadA(A[0], o0, 1); mma(1, A[1], Bf[0]); PWR_W9_SCHED(6)
    loadA(A[1], o1, 1); mma(2, A[2], Bf[0]); PWR_W9_SCHED(6)
    loadA(A[2], o2, 1); loadB(Bf[0], ysn, 0); mma(0, A[0], Bf[1]); PWR_W9_SCHED(8)
    loadA(A[0], o1, 0); mma(1, A[1], Bf[1]); PWR_W9_SCHED(6)      // (event s + 1: its rows are those of the events s - 1, s, s + 1)
    loadA(A[1], o2, 0); mma(2, A[2], Bf[1]); PWR_W9_SCHED(6)
    o0 = o1; o1 = o2; o2 = o3; o3 = o3 + XSLOT == XREGION ? 0 : o3 + XSLOT;
    ys = ysn; ysn = ysn + YSTAGE == NSD * YSTAGE ? 0 : ysn + YSTAGE;
#ifdef PWR_DEBUG_BUILD
    const long long tm0 = p.stamps ? (long long)__builtin_amdgcn_s_memtime() : 0;
#endif
    __syncthreads();                                               // barrier s + 1: this wave's reads of event s have retired; event s + 2 is passed
#ifdef PWR_DEBUG_BUILD
    if (p.stamps) tbm += (long long)__builtin_amdgcn_s_memtime() - tm0;
#endif
  }
#undef PWR_W9_SCHED
#ifdef PWR_DEBUG_BUILD
  if (p.stamps && lane == 0) {
    long long* d = p.stamps + (size_t)blockIdx.x * 32 + wid * 4;
    d[0] = tbm; d[1] = 0; d[2] = (long long)__builtin_amdgcn_s_memtime() - t_loopm; d[3] = nev;
  }
#endif
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      float* __restrict__ out = p.slab + ((size_t)(split * 9 + ky * 3 + kx) * p.CinPad) * p.CoutPad;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ci = ci0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        const int co = co0 + wn * 32 + r;
        out[(size_t)ci * p.CoutPad + co] = acc[ky][kx][e];
      }
    }
}

// 3x3 stride 1, 32-pixel row segments, whole 64-channel tiles on both sides; a split spans at most MAXSB samples' norm states
bool wgrad9w_applicable(const WgradParams& p) {
  const int on = PWR_DBG_ENV("PWR_WGRAD9W", 0);          // (debug build only, read per call: 1 = this kernel instead of conv_wgrad_ws.hip -- measured slower, profiles/r4_experiments.md section 7)
  if (!on || p.ksize != 3 || p.stride != 1 || p.W % 32 || p.M % 32 || p.Cin % 128 || p.Cout % 128 || p.CoutPad != p.Cout || p.CinPad != p.Cin || p.S < 1) return false;
  const int ev_sample = (p.W / 32) * (p.H + 2), total = p.B * ev_sample;
  return (total + p.S - 1) / p.S + 2 <= (w9::MAXSB - 1) * ev_sample;
}

int launch_wgrad9w(const WgradParams& p0, hipStream_t s) {
  WgradParams p = p0;
  p.dbg = PWR_DBG_ENV("PWR_WGRAD9W_DBG", 0);
#ifdef PWR_DEBUG_BUILD
  { const char* e = getenv("PWR_WGRAD3W_STAMPS"); p.stamps = e ? (long long*)strtoull(e, nullptr, 0) : nullptr; }
#endif
  const int ntiles = (p.Cin / 64) * (p.Cout / 64);
  dim3 grid(8 * ntiles * ((p.S + 7) / 8), 1, 1), block(512);
  if (p.in_norm && p.relu_in) hipLaunchKernelGGL((conv_wgrad9w_kernel<true, true>), grid, block, 0, s, p);
  else if (p.in_norm) hipLaunchKernelGGL((conv_wgrad9w_kernel<true, false>), grid, block, 0, s, p);
  else hipLaunchKernelGGL((conv_wgrad9w_kernel<false, false>), grid, block, 0, s, p);
  return (int)hipGetLastError();
}

}  // namespace pwr
   // PWR_DEBUG_BUILD
