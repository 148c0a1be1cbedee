// bf16 weight gradient of a 3x3 stride-1 conv with 128-channel tiles, WAVE-SPECIALISED (round 4): one 512-thread workgroup per CU,
//   waves 0-3  MFMA waves: nothing but transposing LDS fragment reads and v_mfma_f32_32x32x16_bf16 (a 64 ci x 64 co x 3 taps register
//              tile each: 192 accumulator registers -- the 128 x 128 x 3 tile that one wave per SIMD could not feed in rounds 1-3)
//   waves 4-7  loader waves: bring both operands into an LDS ring by LDS-DMA (global_load_lds), apply the operand's pending
//              norm + ReLU to the landed input tile IN LDS, zero the out-of-image halo pixels -- all the VALU work of the layer
//
//   dW[co][ci][ky][kx] = sum_{b, y, x} a[b][y + ky - 1][x + kx - 1][ci] * dy[b][y][x][co]          (/root/reference/model.py:55-63 /
//   :104-112: the weight gradients of the heads' 128 -> 128 convs under autograd; a = the conv's input AFTER its norm + ReLU)
//
// Wave w and wave w + 4 share a SIMD (a workgroup's waves go to SIMDs 0 -> 2 -> 1 -> 3 cyclically), and the matrix pipe and the
// vector ALU of a SIMD are separate pipes: the loader's norm arithmetic (the 78 VALU instructions per wave and K step that the
// 128-wide in-LDS-norm tile of round 3 could not hide behind its own MFMAs) issues beside its partner's MFMAs.  The MFMA waves
// issue no vector-memory instruction at all, so their LDS reads are ordinary compiler-visible loads (the wait-count pass has no
// LDS-DMA in their path to fence with vmcnt(0)): no inline-asm read whose result arrives behind the compiler's back, which is what
// conv_wgrad_dma.hip needs its code-object scan for.  The loader waves' LDS accesses ARE inline asm (they sit behind LDS-DMAs in
// flight) and are SPLIT: tile_read() issues the reads of a landed tile without a wait, tile_wait() is the s_waitcnt lgkmcnt(0) that
// defines their outputs -- the arithmetic of the tile before runs in between.  Nothing the compiler knows orders a use of those outputs
// behind tile_wait(): that is what the code-object scan of the build (codeobj_scan.py: no use of an LDS read in flight) checks for
// this kernel too.
//
// Same decomposition as conv_wgrad3_kernel / conv_wgrad3d_kernel -- workgroup = (split, kernel row ky), K step = 32 output pixels
// of one image row with a 34-pixel input row segment serving kx = 0, 1, 2, per-accumulator MFMA sequence (step by step, K half
// by K half) -- so the slabs are bit-identical to theirs at the same split count, and the same reduce finishes the job.
//
// LDS: a ring of NS = 7 stages (PWR_WS_NS; D = NS - 1 = 6) of [34 px][128 ci] + [32 px][128 co] bf16 rows (17 KiB each, 119 KiB + the norm
// states of up to 8 samples = ~131 KiB per workgroup), 16-byte slots XOR-swizzled by (row & 3) so
// that the four pixel rows of a ds_read_b64_tr_b16 group fall on different bank quarters; the swizzle and the halo go into the
// per-lane SOURCE address of the DMA (cdna_hip_programming.md rule 21).  Per K step s, between barrier s and barrier s + 1:
//   MFMA waves    read stage s (the first fragments were read ahead during step s - 1), 24 MFMAs each
//   loader waves  issue the DMA of step s + D (= s + 6) into the stage read during step s - 1; normalise the tile of step s + 2 in place;
//                 wait until their own pieces of step s + 3 have landed (counted vmcnt: the steps behind it stay in flight)
// so a DMA has D - 3 K steps to land (the ring was 6 stages when the kernel was written; 7 measured 1 - 2 % faster in the step, 8 no
// faster and 3 - 7 % slower isolated: the loop is not latency-bound), a tile is normalised one full step before it is read, and a stage is
// overwritten only after a barrier every MFMA wave reached with its reads retired.
#include <cstdlib>
#include <type_traits>

#include "conv_common.h"
#include "pwr.h"

namespace pwr {

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_ws;

struct WgradPair { WgradParams a, b; };

namespace ws {
constexpr int KP = 32, XROWS = KP + 2, NLW = 4;                  // K step, staged input pixels, loader waves
#ifndef PWR_WS_NS
#define PWR_WS_NS 7
#endif
constexpr int NS = PWR_WS_NS, D = NS - 1;                        // ring stages; a step's pieces are issued D steps ahead
constexpr int MAXSB = 8;                                         // norm states of at most this many samples per split
// geometry for an input tile of BMT channels (128: the heads' convs; 64: the stem's 64 -> 128 conv); the dy tile is always 128 channels wide
template <int BMT> struct Geo {
  static constexpr int RBX = BMT * 2, RBY = 256;                 // bytes per staged pixel
  static constexpr int SPR = RBX / 16;                           // 16-byte slots per input pixel
  static constexpr int XCH = (XROWS * RBX + 1023) / 1024, YCH = KP * RBY / 1024, NCH = XCH + YCH;     // 1-KiB DMA pieces: 9 + 8 / 5 + 8
  static constexpr int NCW = (NCH + NLW - 1) / NLW;              // pieces of loader wave 0 per step (5 / 4); the other waves issue NCW - 1
  static constexpr int XBYTES = XCH * 1024, STAGE = NCH * 1024;
  static constexpr int STATE_BYTES = MAXSB * 3 * BMT * 4;
  static constexpr int LDS_BYTES = NS * STAGE + STATE_BYTES;
};

// byte offset of 16-byte slot `slot` of row `row`: 256-byte rows XOR the slot with (row & 3) << 2, 128-byte rows with ((row >> 1) & 1) << 2
// (conv_wgrad_dma.hip: wswz) -- the four pixel rows of a ds_read_b64_tr_b16 group fall on four different bank quarters either way, and
// both patterns repeat every 4 rows
template <int RB> __device__ __forceinline__ int swzbits(int row) { return RB == 256 ? (row & 3) << 2 : ((row >> 1) & 1) << 2; }
template <int RB> __device__ __forceinline__ int swz(int row, int slot) { return row * RB + ((slot ^ swzbits<RB>(row)) << 4); }
// lane part of the address of an MFMA 32x32x16 operand fragment read by two ds_read_b64_tr_b16 (rows +0 / +4): 8 K values = pixels
// k0 + 8 (lane / 32) .. + 7 of channel chb + lane % 32 (conv_wgrad_dma.hip: wfrag_lane)
template <int RB> __device__ __forceinline__ int frag_lane(int k0, int chb, int lane) {
  const int li = lane & 15, cg = (lane >> 4) & 1, h = lane >> 5, q = li >> 2, pp = li & 3;
  return swz<RB>(k0 + 8 * h + q, (chb >> 3) + 2 * cg + (pp >> 1)) + 8 * (pp & 1);
}
template <int RB> __device__ __forceinline__ bf16x8 frag(const char* a) {
  typedef __attribute__((address_space(3))) bf16x4_ws* lptr;
  const bf16x4_ws lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lptr)a);
  const bf16x4_ws hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lptr)(a + 4 * RB));
  bf16x8 f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3]; f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  return f;
}
constexpr unsigned vmwait(int n) { return (unsigned)((n & 15) | ((n >> 4) << 14) | 0x0070); }      // s_waitcnt vmcnt(n) lgkmcnt(0)
}  // namespace ws

template <bool NRM, bool RELU, int BMT>
__global__ __launch_bounds__(512, 2) void conv_wgrad3w_kernel(WgradPair g) {
  using namespace ws;
  typedef Geo<BMT> G;
  constexpr int RBX = G::RBX, RBY = G::RBY, SPR = G::SPR, XCH = G::XCH, NCH = G::NCH, NCW = G::NCW, XBYTES = G::XBYTES, STAGE = G::STAGE;
  constexpr int LDS_BYTES = G::LDS_BYTES;
  constexpr int MI = BMT / 64;                                   // 32-channel input blocks per MFMA wave (the wave's tile: 32 MI ci x 64 co x 3 taps)
  typedef bf16_t T;
  typedef bf16x8 V;
  const WgradParams p = blockIdx.z ? g.b : g.a;          // (by value: scalar registers, no kernarg reloads inside the loops)
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  // blockIdx.x enumerates (split, ky) so that the three ky workgroups of one split are 8 ids apart (one XCD, speed only)
  const int grp = blockIdx.x / 24, rr = blockIdx.x - grp * 24;
  const int ky = rr >> 3;
  const int split = grp * 8 + (rr & 7);
  if (split >= p.S) return;
  const int ntn = p.CoutPad / 128;
  const int mtile = blockIdx.y / ntn, ntile = blockIdx.y - mtile * ntn;
  const int ci0 = mtile * BMT, co0 = ntile * 128;
  const int step0 = split * p.steps_per_split;
  const int total_steps = p.M / KP;
  int nsteps = total_steps - step0;
  if (nsteps > p.steps_per_split) nsteps = p.steps_per_split;
  const int tiles_x = p.W / KP;
  // tile coordinates of the split's first step
  const int b0 = step0 / (p.H * tiles_x);
  const int rem0 = step0 - b0 * p.H * tiles_x;
  const int y0 = rem0 / tiles_x, x0 = rem0 - y0 * tiles_x;

  if constexpr (NRM) {
    // The norm states (mean, scale, beta) of the samples this split touches go to LDS BEFORE the first DMA is issued: a global load
    // inside the loop would make the wait-count pass put vmcnt(0) -- a wait for every prefetch in flight -- in front of its use.
    const size_t plane = (size_t)p.B * p.Cin;
    float* stl = reinterpret_cast<float*>(smem + NS * STAGE);
    const int lastb = (step0 + (nsteps > 0 ? nsteps - 1 : 0)) / (p.H * tiles_x);
    const int cnt = (lastb - b0 + 1) * 3 * BMT;
    for (int idx = tid; idx < cnt; idx += 512) {
      const int sb = idx / (3 * BMT), k = (idx / BMT) % 3, ch = idx % BMT;
      stl[idx] = p.in_norm[(size_t)(k == 0 ? 0 : k + 1) * plane + (size_t)(b0 + sb) * p.Cin + ci0 + ch];
    }
    __syncthreads();
  }

  if (wid >= NLW) {
    // =================================================================== loader waves
    const int lw = wid - NLW, lt = tid - 64 * NLW;         // loader wave 0..3, loader thread 0..255
    const T* __restrict__ x = reinterpret_cast<const T*>(p.x);
    const T* __restrict__ dy = reinterpret_cast<const T*>(p.dy);
    // per-lane descriptors of this wave's DMA pieces (constant over the steps; only the tile origin moves): loader wave lw issues the
    // pieces lw, lw + 4, ... -- five for wave 0 (it holds the halo pixels 0 and 33, pieces 0 and 8: one clamp delta each), four for the others
    constexpr int I_R = ((XROWS - 1) * RBX / 1024) / NLW;
    static_assert(((XROWS - 1) * RBX / 1024) % NLW == 0 && I_R != 0 && I_R < NCW && NCH == NLW * (NCW - 1) + 1, "wave 0: NCW pieces, the others NCW - 1");
    // (round 5: the pieces go out as RAW BUFFER loads to LDS -- a per-lane 32-bit byte offset that is constant over the steps, the step's
    // tile origin as a scalar 32-bit soffset, the tensor base in a resource descriptor: two instructions per piece.  As 64-bit global
    // pointers every piece cost eight -- a select of the base, a sign extension, a 64-bit vector add -- and the loader waves, which share
    // their SIMD's issue port with an MFMA wave, were the longer side of every K step: ~300 instructions against 24 MFMAs.  Which tensor a
    // piece belongs to is a compile-time property of (piece index, wave 0 or not).  The input resource's base lies one pixel BEFORE the
    // tensor, so that the halo pixel's offset is not negative; tensors below 4 GiB: wgrad3w_applicable.)
    static_assert(XCH % NLW == 1, "piece i of a loader wave is an input piece for every wave != 0 alike");
    int d_lds[NCW], d_voff[NCW], d_dl = 0, d_dr = 0;
#pragma unroll
    for (int i = 0; i < NCW; ++i) {
      int c = lw + NLW * i;
      if (c >= NCH) c = lw;                         // (not issued: only wave 0 has a fifth piece)
      d_lds[i] = c * 1024;
      const bool isx = c < XCH;
      const int cx_ = c < XCH ? c : c - XCH;
      const int q = 64 * cx_ + lane;
      // row and physical slot of this lane's 16 bytes, and the channel slot that belongs there (input tile: SPR slots per pixel; dy: 16)
      const int xr0 = q / SPR, xs1 = q % SPR, yr0 = q >> 4, ys1 = q & 15;
      const int xr = xr0 < XROWS ? xr0 : XROWS - 1;   // rows beyond the 34th: nobody reads them
      const int xoff = xr * p.Cin + ci0 + 8 * (xs1 ^ swzbits<RBX>(xr0));            // (relative to the pixel before the tile's first)
      const int yoff = yr0 * p.Cout + co0 + 8 * (ys1 ^ swzbits<RBY>(yr0));
      d_voff[i] = (isx ? xoff : yoff) * 2;
      if (i == 0) d_dl = (isx && xr == 0) ? p.Cin * 2 : 0;              // out-of-image halo: clamped into the row, zeroed in LDS afterwards
      if (i == I_R) d_dr = (isx && xr == XROWS - 1) ? -p.Cin * 2 : 0;
    }
    const bool five = lw == 0;                          // (wave-uniform)
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(x)) - (size_t)p.Cin * 2, 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(dy), 0, -1, 0x00020000);
    // next step to ISSUE.  NHWC rows are contiguous, so the dy tile of step t starts at pixel 32 t and the input tile of kernel row ky at
    // pixel 32 t + (ky - 1) W (an out-of-image row: the dy tile's own pixels, zero-filled by the pass): one running pixel offset
    int ix = x0, iyy = y0, issued = 0;
    int ipix = step0 * KP;
    const int kyoff = (ky - 1) * p.W;
    auto issue = [&](int soff, auto FIVE) {
      constexpr bool F5 = decltype(FIVE)::value;
      const int iy = iyy + ky - 1;
      const bool rowok = iy >= 0 && iy < p.H;
      const unsigned sx = (unsigned)((ipix + (rowok ? kyoff : 0)) * p.Cin) * 2u, sy = (unsigned)(ipix * p.Cout) * 2u;
      const int first = ix == 0 ? d_dl : 0, last = ix == tiles_x - 1 ? d_dr : 0;
      typedef __attribute__((address_space(3))) void* ldsp;
      char* base = smem + soff;
#pragma unroll
      for (int i = 0; i < NCW; ++i) {
        if (i == NCW - 1 && !F5) break;
        constexpr int XP = (XCH + NLW - 1) / NLW;         // wave 0 has XP input pieces, the others XP - 1
        const bool isx = i < (F5 ? XP : XP - 1);
        const int vo = d_voff[i] + (i == 0 ? first : 0) + (i == I_R ? last : 0);
        if (isx) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (ldsp)(base + d_lds[i]), 16, vo, (int)sx, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsy, (ldsp)(base + d_lds[i]), 16, vo, (int)sy, 0, 0);
      }
      ++issued;
      ipix += KP;
      if (++ix == tiles_x) { ix = 0; if (++iyy == p.H) iyy = 0; }
    };
    // ---- the in-LDS pass over a landed input tile: norm + ReLU (NRM), zeros for the out-of-image halo pixels, and an all-zero tile where
    // the whole input ROW lies outside the image (ky = 0 / 2 at the top / bottom: the step then adds nothing, without a branch or a select
    // in the MFMA waves).  Loader thread lt owns the 16-byte slots lt and lt + 256 of the tile's 34 x 16 slots (rows lt / 16 and + 16); the
    // 32 slots of rows 32 and 33 go to the threads 192 .. 223 (rows 12, 13 + 20: the SAME eight channels, the swizzle repeats every 4 rows)
    typedef __attribute__((address_space(3))) char* lds_ptr;
    const unsigned lds0 = (unsigned)(size_t)(lds_ptr)smem;
    const unsigned nrl = lds0 + lt * 16;
    // (BMT = 64: the tile has 34 x 8 = 272 slots: slot lt, and the 16 slots of rows 32, 33 for the threads 192 .. 207 = rows 24, 25 + 8)
    const int nr_row = lt / SPR;
    const int nr_ch = 8 * ((lt % SPR) ^ swzbits<RBX>(nr_row));     // first of the eight channels (relative to ci0)
    constexpr int NEXTRA = 2 * SPR;                                // slots of the rows 32 and 33
    const bool third = lt >= 192 && lt < 192 + NEXTRA;            // (wave 3 only)
    constexpr int OFF3 = (32 * SPR - 192) * 16;                    // byte offset of the third slot from the first
    constexpr bool TWO = BMT == 128;                              // a second main slot (lt + 256) exists
    float mu[8], sc[8], be[8];
    int state_b = -1;
    int nb = b0, ny = y0, nx = x0;                                // tile coordinates of the step whose tile is processed next
    auto nr_state = [&]() {                                       // (a change of sample: at most every H * W / 32 steps)
      if (nb != state_b) {
        const unsigned a = lds0 + NS * STAGE + (nb - b0) * (3 * BMT * 4) + nr_ch * 4;
        f32x4 q0, q1, q2, q3, q4, q5;
        asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:16\n\tds_read_b128 %2, %6 offset:%7\n\tds_read_b128 %3, %6 offset:%8\n\t"
                     "ds_read_b128 %4, %6 offset:%9\n\tds_read_b128 %5, %6 offset:%10\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&v"(q4), "=&v"(q5)
                     : "v"(a), "n"(BMT * 4), "n"(BMT * 4 + 16), "n"(2 * BMT * 4), "n"(2 * BMT * 4 + 16) : "memory");
#pragma unroll
        for (int e = 0; e < 4; ++e) { mu[e] = q0[e]; mu[4 + e] = q1[e]; sc[e] = q2[e]; sc[4 + e] = q3[e]; be[e] = q4[e]; be[4 + e] = q5[e]; }
        state_b = nb;
      }
    };
    auto nr_math = [&](f32x4 raw) {                               // conv_wgrad3_kernel's arithmetic: bit-identical tile
      // (measured and dropped: the ReLU on the rounded values as packed int16 max -- v_pk_max_i16, 8 instead of 12 vector instructions per
      // eight channels, bit-identical -- made the kernel 10 % SLOWER isolated: 83 against 75.5 us)
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
    // The pass is split in two so that the LDS round trip of the raw values hides behind the arithmetic of the PREVIOUS tile (the serial
    // chain read -> wait -> arithmetic -> store -> wait per K step was longer than the MFMA waves' 24 MFMAs: elimination in
    // profiles/r4_experiments.md section 1):  tile_read() issues the reads of a landed tile (inline asm, no wait), tile_wait() is the
    // lgkmcnt wait TIED to the three registers (every use depends on it), tile_finish() computes and stores.  The values cross one
    // barrier in registers.  All loader waves read three slots (the third lies inside the stage for every thread; only the threads
    // 192 .. 223 use it): one straight-line asm per step, no merge of an asynchronous register with a compiler-written one.
    auto tile_read = [&](int soff, f32x4& r0, f32x4& r1, f32x4& r2) {
      if constexpr (NRM) {
        const unsigned a = nrl + soff;
        asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:4096\n\tds_read_b128 %2, %3 offset:%4"
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2) : "v"(a), "n"(OFF3) : "memory");
      }
    };
    auto tile_wait = [&](f32x4& r0, f32x4& r1, f32x4& r2) {
      if constexpr (NRM) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2)::"memory");
    };
    auto tile_finish = [&](int soff, const f32x4& r0, const f32x4& r1, const f32x4& r2) {      // soff: byte offset of the ring stage that holds the tile
      const bool zl = nx == 0, zr = nx == tiles_x - 1;            // the tile touches the left / right image border
      const int iy = ny + ky - 1;
      const bool rowzero = iy < 0 || iy >= p.H;                   // the whole input row is zero padding
      const unsigned a = nrl + soff;
      if constexpr (NRM) {
        nr_state();
        f32x4 o0, o1;
        o0 = nr_math(r0); o1 = TWO ? nr_math(r1) : r1;
        if (rowzero | zl | zr) {                                   // (wave-uniform, rare: image borders)
          if (rowzero || (zl && nr_row == 0)) o0 = zero4;
          if (rowzero) o1 = zero4;
        }
        // (the debug build's elimination switches for this pass -- raw values stored back, arithmetic without stores: round 4 -- are gone: their
        // out-of-line blocks sat textually behind an LDS read in flight and tripped the build's linear code-object scan)
        {
          asm volatile("ds_write_b128 %0, %1" ::"v"(a), "v"(o0) : "memory");
          if constexpr (TWO) asm volatile("ds_write_b128 %0, %1 offset:4096" ::"v"(a), "v"(o1) : "memory");
        }
        if (lw == NLW - 1) {                                       // (wave-uniform: the wave that owns rows 32 and 33)
          f32x4 o2 = nr_math(r2);
          if (rowzero || (zr && lt >= 192 + SPR)) o2 = zero4;      // (the upper half of the extra threads holds row 33)
          if (third) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(a), "v"(o2), "n"(OFF3) : "memory");
        }
      } else {
        if (rowzero) {
          asm volatile("ds_write_b128 %0, %1" ::"v"(a), "v"(zero4) : "memory");
          if constexpr (TWO) asm volatile("ds_write_b128 %0, %1 offset:4096" ::"v"(a), "v"(zero4) : "memory");
          if (third) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(a), "v"(zero4), "n"(OFF3) : "memory");
        } else if (lt < SPR) {                                     // one pixel row = SPR slots
          if (zl) asm volatile("ds_write_b128 %0, %1" ::"v"(a), "v"(zero4) : "memory");
          if (zr) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(a), "v"(zero4), "n"((XROWS - 1) * RBX) : "memory");
        }
      }
      if (++nx == tiles_x) { nx = 0; if (++ny == p.H) { ny = 0; ++nb; } }
    };

    // ---- one copy of prologue + loop per kind of loader wave (wave 0: five DMA pieces per step; wave 3: the rows 32, 33 of the tile pass),
    // so that the piece count, the wait counts and the extra slot are compile-time in the loop -- and the loop in two parts: the STEADY
    // part (every step issues, finishes a tile and waits for the same count: no conditional but the rare image-border ones) and the
    // last D steps (general).  The loop's own scalar control flow was 0.3 us of every 0.6 - 0.8 us K step before (elimination,
    // profiles/r4_experiments.md section 7): the MFMA waves wait for it in the barrier.
    auto run_loader = [&](auto FIVE) {
      constexpr int NPW = decltype(FIVE)::value ? NCW : NCW - 1;       // this wave's DMA pieces per step
      auto landed = [&](int k) {      // all but this wave's pieces of the k most recently issued steps have landed; every LDS access retired
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        static_assert(D - 3 <= 5 && 5 * NCW < 64, "wait cases below; vmcnt is a 6-bit counter");
        if (k >= 5) __builtin_amdgcn_s_waitcnt(vmwait(5 * NPW));
        else if (k == 4) __builtin_amdgcn_s_waitcnt(vmwait(4 * NPW));
        else if (k == 3) __builtin_amdgcn_s_waitcnt(vmwait(3 * NPW));
        else if (k == 2) __builtin_amdgcn_s_waitcnt(vmwait(2 * NPW));
        else if (k == 1) __builtin_amdgcn_s_waitcnt(vmwait(NPW));
        else __builtin_amdgcn_s_waitcnt(vmwait(0));
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
      };
      // prologue: steps 0 .. D-1 in flight, tiles 0 and 1 finished, the raw tile 2 in registers, tile 3 landed
#pragma unroll
      for (int k = 0; k < D; ++k)
        if (k < nsteps) issue(k * STAGE, FIVE);
      landed(issued - 3);                                            // tiles 0, 1 and 2 are needed
      __builtin_amdgcn_s_barrier();                                  // (P) every loader's pieces of tiles 0 .. 2 have landed
      __atomic_signal_fence(__ATOMIC_SEQ_CST);
      f32x4 rc0 = zero4, rc1 = zero4, rc2 = zero4, rn0, rn1, rn2;
      {
        f32x4 t0, t1, t2;
        tile_read(0, t0, t1, t2); tile_wait(t0, t1, t2); tile_finish(0, t0, t1, t2);
        if (nsteps > 1) { tile_read(STAGE, t0, t1, t2); tile_wait(t0, t1, t2); tile_finish(STAGE, t0, t1, t2); }
      }
      tile_read(2 * STAGE, rc0, rc1, rc2);
      tile_wait(rc0, rc1, rc2);
      landed(issued - 4);
      int stg = 0, s = 0;                                            // ring stage of step s
      // steady part: s + D < nsteps, so step s + D is issued, tile s + 2 finished, and exactly D - 4 later steps stay in flight
      // (two steps per trip, the raw tile alternating between two register sets: a copy of twelve registers per step was six of the step's
      // ~250 instructions, and every instruction of this loop is issue time the MFMA waves wait for)
      auto steady = [&](f32x4& c0, f32x4& c1, f32x4& c2, f32x4& n0, f32x4& n1, f32x4& n2) __attribute__((always_inline)) {
        __builtin_amdgcn_s_barrier();                                // barrier s: every loader's pieces of tile s + 3 have landed
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        const int s2 = stg + 2 >= NS ? stg + 2 - NS : stg + 2, s3 = stg + 3 >= NS ? stg + 3 - NS : stg + 3;
        tile_read(s3 * STAGE, n0, n1, n2);
        const int istg = stg == 0 ? NS - 1 : stg - 1;                // (s + D) % NS: the stage read during step s - 1
#ifdef PWR_DEBUG_BUILD
        if (!(p.dbg & 16))                                           // (elimination: no DMA after the prologue)
#endif
        issue(istg * STAGE, FIVE);
        tile_finish(s2 * STAGE, c0, c1, c2);
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        __builtin_amdgcn_s_waitcnt(vmwait((D - 4) * NPW));           // step s + 4 landed (own pieces), every LDS access retired
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        tile_wait(n0, n1, n2);
        stg = stg + 1 == NS ? 0 : stg + 1;
      };
#pragma nounroll
      for (; s + D + 1 < nsteps; s += 2) {
        steady(rc0, rc1, rc2, rn0, rn1, rn2);
        steady(rn0, rn1, rn2, rc0, rc1, rc2);
      }
      if (s + D < nsteps) {
        steady(rc0, rc1, rc2, rn0, rn1, rn2);
        rc0 = rn0; rc1 = rn1; rc2 = rn2;
        ++s;
      }
      // the last D steps: nothing left to issue, the pipeline drains
#pragma nounroll
      for (; s < nsteps; ++s) {
        __builtin_amdgcn_s_barrier();
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        const int s2 = stg + 2 >= NS ? stg + 2 - NS : stg + 2, s3 = stg + 3 >= NS ? stg + 3 - NS : stg + 3;
        tile_read(s3 * STAGE, rn0, rn1, rn2);                        // (past the last step: a stage nobody uses any more)
        if (s + 2 < nsteps) tile_finish(s2 * STAGE, rc0, rc1, rc2);
        landed(nsteps - 1 - (s + 4));
        tile_wait(rn0, rn1, rn2);
        rc0 = rn0; rc1 = rn1; rc2 = rn2;
        stg = stg + 1 == NS ? 0 : stg + 1;
      }
    };
    if (five) run_loader(std::true_type{});
    else run_loader(std::false_type{});
    __builtin_amdgcn_s_waitcnt(vmwait(0));                         // (no DMA may be in flight when the workgroup's LDS is released)
    __builtin_amdgcn_s_barrier();                                  // barrier nsteps (the MFMA waves' last)
    return;
  }

  // ===================================================================== MFMA waves
  const int wm = wid >> 1, wn = wid & 1;
  f32x16 acc[3][MI][2];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][i][j][e] = 0.f;
  // lane parts of the fragment addresses: input tile per (tap, 32-channel block), dy tile per 32-channel block
  const char* xl[3][MI];
  const char* yl[2];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int i = 0; i < MI; ++i) xl[t][i] = smem + frag_lane<RBX>(t, wm * (BMT / 2) + i * 32, lane);
#pragma unroll
  for (int j = 0; j < 2; ++j) yl[j] = smem + XBYTES + frag_lane<RBY>(0, wn * 64 + j * 32, lane);

  // unit u = (K half h = u / 3, tap t = u % 3): two input fragments (the wave's two 32-channel blocks), four MFMAs against the two dy
  // fragments of half h.  Input fragments in a ring of three units (read two units ahead), dy fragments double-buffered by half.
  V A[3][MI], Bf[2][2];
  auto loadA = [&](V (&a)[MI], int soff, int u) {
    const int h = u / 3, t = u - 3 * h;
#pragma unroll
    for (int i = 0; i < MI; ++i) a[i] = frag<RBX>(xl[t][i] + soff + h * 16 * RBX);
  };
  auto loadB = [&](V (&b)[2], int soff, int h) {
    b[0] = frag<RBY>(yl[0] + soff + h * 16 * RBY);
    b[1] = frag<RBY>(yl[1] + soff + h * 16 * RBY);
  };
  auto mma = [&](int t, const V (&a)[MI], const V (&b)[2]) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[t][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[t][i][j], 0, 0, 0);
  };
  // interleave: one MFMA, then the next unit's LDS reads behind it (a unit = 2 MI MFMAs; its reads: 2 MI for the input fragments, + 4 where
  // a dy pair is read)
  constexpr int RA = 2 * MI, RAB = 2 * MI + 4, MREST = 2 * MI - 1;
#define PWR_WS_SCHED(reads)                                  \
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       \
  __builtin_amdgcn_sched_group_barrier(0x100, reads, 0);   \
  __builtin_amdgcn_sched_group_barrier(0x008, MREST, 0);

  __syncthreads();                                                 // (P)
  __syncthreads();                                                 // barrier 0: stages 0 and 1 are complete
  loadA(A[0], 0, 0); loadA(A[1], 0, 1); loadB(Bf[0], 0, 0);
  int stg = 0;
#pragma nounroll
  for (int s = 0; s < nsteps; ++s) {
#ifdef PWR_DEBUG_BUILD
    if (p.dbg & 8) { __syncthreads(); continue; }                  // (elimination: the MFMA waves only keep the barriers)
#endif
    const int soff = stg * STAGE;
    const int nstg = stg + 1 == NS ? 0 : stg + 1;
    const int noff = nstg * STAGE;
    // (an out-of-image input ROW -- ky = 0 / 2 at the top / bottom -- arrives as an all-zero tile from the loader waves: acc + 0 = acc bit
    // for bit, one straight-line loop body; a branch around the MFMAs made the register allocator copy the 192 accumulators between the
    // two paths and spill them)
    loadA(A[2], soff, 2); loadB(Bf[1], soff, 1);
    mma(0, A[0], Bf[0]); PWR_WS_SCHED(RAB)
    loadA(A[0], soff, 3); mma(1, A[1], Bf[0]); PWR_WS_SCHED(RA)
    loadA(A[1], soff, 4); mma(2, A[2], Bf[0]); PWR_WS_SCHED(RA)
    loadA(A[2], soff, 5); loadB(Bf[0], noff, 0);
    mma(0, A[0], Bf[1]); PWR_WS_SCHED(RAB)
    loadA(A[0], noff, 0); mma(1, A[1], Bf[1]); PWR_WS_SCHED(RA)
    loadA(A[1], noff, 1); mma(2, A[2], Bf[1]); PWR_WS_SCHED(RA)
    stg = nstg;
    __syncthreads();                                               // barrier s + 1: this wave's reads of stage s have retired; stage s + 2 is complete
  }
#undef PWR_WS_SCHED
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    float* __restrict__ out = p.slab + ((size_t)(split * 9 + ky * 3 + t) * p.CinPad) * p.CoutPad;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int ci = ci0 + wm * (BMT / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          const int co = co0 + wn * 64 + j * 32 + r;
          out[(size_t)ci * p.CoutPad + co] = acc[t][i][j][e];
        }
  }
}

// 3x3 stride 1, 32-pixel row segments, whole 128-channel output tiles, input channels a multiple of 128 (128-channel input tiles; the
// kernel is also generic over a 64-channel input tile -- the stem's 64 -> 128 conv, model.py:174 -- which the debug build can select); a
// split spans at most MAXSB samples' norm states
bool wgrad3w_applicable(const WgradParams& p) {
  // (debug build: read per call, so that one process can A/B the kernels; 3 = also the 64-channel input tile -- the stem's 64 -> 128 conv:
  // measured no gain in the train step, 5.81 - 5.84 ms with 48 - 96 splits against 5.81 - 5.83 on the register-staged kernel, so the
  // shipped library takes 128-channel input tiles only)
  const int on = PWR_DBG_ENV("PWR_WGRAD3W", 1);
  const bool cin_ok = p.Cin % 128 == 0 ? p.CinPad == p.Cin : (p.Cin == 64 && on == 3);
  if (!on || p.ksize != 3 || p.stride != 1 || p.W % 32 || p.M % 32 || !cin_ok || p.Cout % 128 || p.CoutPad != p.Cout) return false;
  if (((long long)p.M + 2 * p.W) * (p.Cin > p.Cout ? p.Cin : p.Cout) * 2 >= (1ll << 32)) return false;      // (32-bit byte offsets of the buffer loads)
  if (on == 2 && p.in_norm) return false;      // (2: only the layers whose operand carries no norm; the norm-fed ones take the register-staged kernel)
  return p.steps_per_split <= (ws::MAXSB - 1) * (p.H * p.W / 32);
}

// one job (b == nullptr) or two jobs of ONE geometry (the two regression heads' layers of the same depth) in one launch
int launch_wgrad3w(const WgradParams& a, const WgradParams* b, hipStream_t s) {
  if (b && (a.B != b->B || a.H != b->H || a.W != b->W || a.Cin != b->Cin || a.Cout != b->Cout || a.S != b->S || a.steps_per_split != b->steps_per_split ||
            (a.in_norm == nullptr) != (b->in_norm == nullptr)))
    return PWR_EINVAL;
  WgradPair g{a, b ? *b : a};
  g.a.dbg = g.b.dbg = PWR_DBG_ENV("PWR_WGRAD3W_DBG", 0);     // (debug build: timing by elimination, results are WRONG)
  const int bmt = a.Cin % 128 == 0 ? 128 : 64;
  dim3 grid(24 * ((a.S + 7) / 8), (a.Cin / bmt) * (a.Cout / 128), b ? 2 : 1), block(512);
  if (bmt == 128) {
    if (a.in_norm && a.relu_in) hipLaunchKernelGGL((conv_wgrad3w_kernel<true, true, 128>), grid, block, 0, s, g);
    else if (a.in_norm) hipLaunchKernelGGL((conv_wgrad3w_kernel<true, false, 128>), grid, block, 0, s, g);
    else hipLaunchKernelGGL((conv_wgrad3w_kernel<false, false, 128>), grid, block, 0, s, g);
  } else {
    if (a.in_norm && a.relu_in) hipLaunchKernelGGL((conv_wgrad3w_kernel<true, true, 64>), grid, block, 0, s, g);
    else if (a.in_norm) hipLaunchKernelGGL((conv_wgrad3w_kernel<true, false, 64>), grid, block, 0, s, g);
    else hipLaunchKernelGGL((conv_wgrad3w_kernel<false, false, 64>), grid, block, 0, s, g);
  }
  return (int)hipGetLastError();
}

}  // namespace pwr
