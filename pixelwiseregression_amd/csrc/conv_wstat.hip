// 3x3 stride-1 convolution 128 -> 128 channels, bf16, WEIGHT-STATIONARY and persistent (round 5): the heads' convs
// (/root/reference/model.py:54-65 / :103-114: 72 % of the conv FLOPs of a forward pass), plain, with the norm + ReLU prologue, with the
// forward-statistics epilogue and with the norm-backward-sums epilogue of their data gradients.
//
// Why another form.  conv3x3_patch_kernel (conv_patch.hip) spends 30 % of a workgroup's life staging its input patch, 53 % in the K loop and
// 17 % in the epilogue, the two workgroups of a CU run in phase, and it streams all 288 KiB of weights through LDS for every 128-pixel tile
// (a 3-stage LDS-DMA ring with a barrier per 32-channel K step): 0.355 of the bf16 MFMA peak for two rounds.  Here:
//   * ONE workgroup of 4 waves per CU, ONE wave per SIMD, all 512 registers of the SIMD; the workgroup is PERSISTENT and walks a contiguous
//     range of 4 x 32-pixel tiles.
//   * The WEIGHTS live in REGISTERS for the whole launch: wave w owns output channels 32 w ... 32 w + 31 for the whole K extent
//     (9 taps x 128 input channels = 72 K steps of 16 channels x 4 registers = 288 registers: 256 AGPRs + 32 VGPRs), loaded once from L2 --
//     from a pack in the kernel's own fragment order every fragment is one contiguous KiB (conv_mfma.hip PackDesc::order).  No weight
//     traffic through LDS, no weight ring, NO barrier inside the K loop; LDS read traffic per MFMA halves (only pixels are read from LDS).
//   * MFMA operands are swapped against conv_patch.hip: A = weights (rows = output channels), B = pixels (columns = one tile row),
//     v_mfma_f32_32x32x16_bf16, accumulators in VGPRs.  With the channel <-> MFMA-row assignment of the weight load a lane's 16 accumulator
//     registers of a tile row are 16 CONSECUTIVE channels of one pixel: the epilogue is bias + round + two 16-byte NHWC stores per row
//     straight from registers (no LDS round trip).  Same K-step order and same operands as the patch kernel's 16x16x32 form => outputs
//     bit-identical to it; the epilogue statistics keep its summation order through one v_permlane16_swap per packed dword (below):
//     tests/test_00_kat_gpu.py holds both kernels to the same digests, tools/wstat_check.py compares them directly.
//   * The input patch (6 x 34 pixels x 128 channels, norm + ReLU applied on the way, pixel pitch padded by 16 B like conv_patch.hip) is DOUBLE
//     buffered in LDS (2 x 55 KiB): the patch of tile n + 1 is loaded, normalised and written while the MFMAs of tile n issue -- the wave's own
//     software pipeline, since a second wave per SIMD does not fit beside 288 weight registers.  One barrier per tile.
//   * A tile runs as two half-tile K loops (tile rows 0 - 1, then 2 - 3); the epilogue of the half finished before rides behind the MFMAs
//     of the running one.  ALL vector work inside the K loops -- staging and epilogue -- is cut into ITEMS of one or two independent vector
//     instructions that are handed out per MFMA slot by a compile-time table (wst::make_deal) under a budget of three extra instructions
//     per slot; a full scheduling barrier closes every slot and an empty asm ties each slot's temporaries into it (left alone hipcc emits a
//     vector's 40 staging instructions as one lump and the pipe drains).  Why three: with ONE wave per SIMD everything issues from the wave
//     that issues the MFMAs, one instruction per four cycles; tools/probes/issue_probe.cpp measured a slot of MFMA + fragment read + wait
//     at 32.5 cycles, 33.5 / 34 / 35 with one / two / three more instructions, +4 ... 5 for each one beyond, 8 for a scalar instruction
//     beyond two, 22 for a v_pk_add_f32 (profiles/r5_issue_probe.jsonl).
//   * Addresses are raw buffer loads / stores: a scalar 32-bit row offset in soffset, a per-thread 32-bit offset, the tensor base in a
//     resource descriptor -- as 64-bit pointers the row bases of a tile were ~50 scalar instructions between the barrier and the tile's
//     first MFMA.  The zero padding is folded into the ReLU: v_med3_f32(x, 0, keep), keep = +inf or 0.
//   * The weights of taps 3 .. 7 are requested from INSIDE the first tile's first K loop, three taps ahead of their use (vmcnt counts in
//     order: requested up front, the first wait for anything behind them is a wait for all 288 KiB).  They still bound the first tile:
//     inside this kernel a CU gets ~26 B / clock of them while all 32 CUs of the XCD fetch the same 288 KiB (~11 k cycles in either arrangement;
//     not the L2's limit: tools/probes/wload_probe.cpp loads them alone at 55 B / clock per CU, in any order).
//   * KIND 3, the NARROW form (the heads' last conv, 128 -> J <= 32, fp32 NCHW out): the four waves hold the same 32 channels and split the
//     tile's rows; the staging pipeline is two tiles deep there (the registers exist).  Staging-bound: 13 x 38 vector instructions per tile
//     beside 72 MFMAs.
// Work per tile and wave: 288 MFMAs (9216 matrix-pipe cycles), 288 ds_read_b128, 13 loads + 13 ds_write_b128 + ~490 vector instructions of
// staging (norm form), ~100 (plain) / ~400 (forward statistics) of epilogue.  Measured (MI355X, B = 32, 64 x 64, s_memtime stamps):
// prologue 10 k cycles, the first tile's first K loop 7.9 k (waiting for weights), the others 5.2 - 5.7 k (MFMA alone: 4.7 k; by
// elimination the staging costs ~0.45 k per half, the epilogue ~0.15 k).  The norm-backward-sums form (KIND 2, debug build only) runs
// 9 - 11 k: its 9 vector instructions per element do not fit beside one wave's MFMAs, so the data gradients stay on conv_patch.hip's pair
// kernel (two workgroups per CU hide that epilogue better).
#include <type_traits>
#include <utility>

#include "conv_common.h"
#include "pwr.h"

namespace pwr {

struct WstatArgs {
  ConvParams job[2];
  int njobs, wgs_per_job;
  int tiles_q, tiles_r;      // tiles / wgs_per_job and the remainder: workgroup c of a job takes tiles [c q + min(c, r), ... + q + (c < r))
  // (everything a workgroup would have to DIVIDE for, precomputed: eight integer divisions were ~150 of the ~250 scalar instructions a
  // workgroup issued before its first load)
  int job_shift, nx_shift;   // njobs = 1 << job_shift; the workgroups of a job sit on nx = 8 >> job_shift XCD labels, nx = 1 << nx_shift
  int xq, xr;                // wgs_per_job / nx and the remainder
  unsigned magic_img, magic_x;   // ceil(2^32 / tiles_img), ceil(2^32 / tiles_x): n / d = mulhi(n, magic), one too large at most (fixed up)
};

#ifdef PWR_DEBUG_BUILD
long long* wstat_stamps();        // conv_patch.hip: the buffer of pwr_debug_set_stamps
#endif

// (debug build variants, tools/build_debug.py -DWST_DBG=bits: timing by elimination, results WRONG -- 1 no staging in the K loops, 2 no epilogue
// in the K loops, 4 no fragment reads in the K loops, 8 no MFMAs, 16 (KIND 2) the y loads of the statistics from one cache-hot row)
#ifndef WST_DBG
#define WST_DBG 0
#endif
#ifndef WST_AGPR_STEPS
#define WST_AGPR_STEPS 32
#endif
#ifndef WST_RING
#define WST_RING 6
#endif
#ifndef WST_TAPS_AHEAD
#define WST_TAPS_AHEAD(NRM, KIND) (((KIND) == 2 || ((KIND) == 1 && (NRM))) ? 9 : 3)
#endif

namespace wst {
constexpr int PW = 34, PP = 6 * 34, COUT = 128;
// Everything that depends on the INPUT channel count CI (128: the heads' convs; 64: the stem's 64 -> 128 conv, model.py:174-176): K steps
// of 32 channels per tap, the patch pixel's pitch (padded by 16 B), the staging vectors per thread -- 204 pixels x CI / 8 slots / 256
// threads: 13 (a buffer holds 208 pixels: the last round's surplus pixels are written, never read, no branch) or 7 (224 pixels).
// A thread stages one 16-byte channel slot of PXR consecutive pixels' worth per round: vectors k < KX are patch row k / VPR, columns
// PXR (k % VPR) + st_pl; vector KX is the patch's two right columns.
template <int CI> struct Shape {
  static_assert(CI == 64 || CI == 128, "input channels");
  static constexpr int KCH = CI / 32, ITERS = 9 * KCH, PITCH = CI * 2 + 16, SPP = CI / 8, PXR = 256 / SPP, VPR = 32 / PXR, KX = 6 * VPR, NITP = KX + 1;
  static constexpr int PATCH_BYTES = NITP * PXR * PITCH;
  static constexpr int SLOTS = ITERS * 8, HSLOTS = ITERS * 4, NSLOTS = ITERS * 2;
};
constexpr int MAX_SLOTS = Shape<128>::SLOTS, MAX_NITP = Shape<128>::NITP;
// Schedule inside the K loops.  A SLOT is one v_mfma_f32_32x32x16_bf16 (288 per tile, 72 in the narrow form; 32 matrix-pipe cycles).  With ONE
// wave per SIMD the wave that issues the MFMAs issues everything else too, one instruction per four cycles, and the MFMA itself occupies
// the issue port for a while: measured (tools/probes/issue_probe.cpp, profiles/r5_issue_probe.jsonl) a slot of MFMA + fragment read +
// its wait runs in 32.5 cycles, with 1 / 2 / 3 more instructions of ANY kind (vector, s_nop) in 33.5 / 34 / 35, and every further one
// costs 4 - 5 (39, 44, 49); a scalar instruction beyond two costs 8, a v_pk_add_f32 22.  So what matters is that NO slot carries more
// than three extra instructions: the staging of the next patch (13 vectors x SV items) and the epilogue of the finished half tile are
// ITEMS with an instruction cost, dealt out to the slots by the compile-time table below -- the epilogue proportionally over its half, the
// staging in order over what the epilogue leaves of three instructions per slot -- instead of by item count.
constexpr int S_START = 40, LEAD = 40;
// staging items of a vector.  Norm form, per channel pair j (i = 5 j + o): {unpack lo, unpack hi (+ the vector's keep value, i = 0)}
// {- mean x 2} {fma x 2} {ReLU-and-mask x 2: v_med3_f32(x, 0, keep), keep = +inf or 0} {round + pack}; i = 20: the 16-byte LDS store.
// Plain form: i = 0 .. 3 the mask of one dword, i = 4 the store.
__host__ __device__ constexpr int sv_items(bool nrm) { return nrm ? 21 : 5; }
__host__ __device__ constexpr int s_cost(bool nrm, bool nar, int i) {
  if (i == sv_items(nrm) - 1) return nar ? 2 : 1;     // (narrow form: the LDS store and the global load of the vector's next-but-one tile)
  if (!nrm) return i == 0 ? 2 : 1;
  return i == 0 ? 3 : (i % 5 == 4 ? 1 : 2);
}
// epilogue items per tile row (see epi_micro): 14 base items (6 x two vector instructions, a store) per 16 channels, then the statistics
__host__ __device__ constexpr int e_st(int kind) { return kind == 1 ? 34 : (kind == 2 ? 2 + 72 : 0); }
// (the shift of the forward statistics: 128 input channels -- the ROUNDED output at the tile's first pixel, like the patch kernel's one-pass
// epilogue; 64 -- its fp32 value, like the two-pass epilogue of the patch kernel form that layer ran on: the sums are bit-identical to
// the kernel they replace in both cases)
__host__ __device__ constexpr int e_a0(int kind, int ci = 128) { return kind == 1 ? (ci == 64 ? 8 : 6) : 0; }
__host__ __device__ constexpr int e_bf(int kind, int ci = 128) { return (kind == 1 || kind == 2) ? (ci == 64 ? 40 : 48) : 0; }
__host__ __device__ constexpr int e_row(int kind) { return 14 + e_st(kind); }
__host__ __device__ constexpr int e_half(int kind, int half, int ci = 128) { return kind == 3 ? 0 : 2 * e_row(kind) + (half ? e_a0(kind, ci) : e_bf(kind, ci)); }
__host__ __device__ constexpr int e_cost(int kind, int half, int eu, int ci = 128) {
  const int r0n = e_row(kind) + (half ? e_a0(kind, ci) : 0);
  const int u = eu < r0n ? eu : (eu - r0n < e_row(kind) ? eu - r0n : 99);
  if (ci == 64 && half && eu >= 16 && eu < 16 + e_a0(kind, ci)) return 6;      // (an fp32 shift item: two adds, two lane reads, a select)
  return (u < 14 && u % 7 == 6) ? 3 : 2;             // (a store: its scalar address arithmetic rides with it)
}
struct Deal {
  short s_lo[MAX_SLOTS + 1];    // staging items [s_lo[g], s_lo[g + 1]) run in slot g (item m = vector m / SV, step m % SV)
  short e_lo[MAX_SLOTS + 1];    // epilogue items likewise; items >= e_half(kind, 0) belong to the second half (index - e_half(kind, 0))
  signed char ld[MAX_SLOTS];    // the vector whose global load is issued in slot g, or -1
  short max_s, max_e, max_cost; // (for the static_asserts and the fixed-trip loops of the kernel)
};
__host__ __device__ constexpr Deal make_deal(bool nrm, int kind, int ci = 128) {
  Deal d{};
  const bool nar = kind == 3;
  const int ITERS = 9 * (ci / 32), SLOTS = ITERS * 8, HSLOTS = ITERS * 4, NSLOTS = ITERS * 2, NITP = ci == 128 ? 13 : 7;
  const int nslots = nar ? NSLOTS : SLOTS;
  const int sv = sv_items(nrm), ns = NITP * sv;
  int ecost[MAX_SLOTS] = {};
  const int eha = e_half(kind, 0, ci);
  for (int half = 0; half < 2 && !nar; ++half) {
    const int eh = e_half(kind, half, ci);
    int ce = 0;
    for (int i = 0; i < eh; ++i) ce += e_cost(kind, half, i, ci);
    int idx = 0, cum = 0;
    for (int sl = 0; sl < HSLOTS; ++sl) {
      const int g = half * HSLOTS + sl;
      d.e_lo[g] = (short)((half ? eha : 0) + idx);
      while (idx < eh && (2 * cum + e_cost(kind, half, idx, ci)) * HSLOTS < 2 * ce * (sl + 1)) {     // the item's midpoint falls into this slot
        ecost[g] += e_cost(kind, half, idx, ci); cum += e_cost(kind, half, idx, ci); ++idx;
      }
    }
  }
  for (int g = nar ? 0 : SLOTS; g <= MAX_SLOTS; ++g) d.e_lo[g] = (short)(nar ? 0 : eha + e_half(kind, 1, ci));
  // staging: in order, proportionally over the slots from s_start on, never lifting a slot above `cap` instructions; cap = the smallest that
  // fits.  (Wide forms: a vector's global load goes out LEAD slots before its first item and the first S_START slots of a tile carry no
  // staging, so that even the first vector's load -- issued in slot 0, from L2 / the Infinity Cache -- has ~1300 cycles to land.  The
  // narrow form has the registers for a pipeline TWO tiles deep: a vector's load for the tile after next rides with its store item.)
  const int s_start = nar ? 0 : S_START, margin = nar ? 2 : 8;      // (aim `margin` slots short of the end: a slot that is full makes an item wait)
  int cs = 0;
  for (int m = 0; m < ns; ++m) cs += s_cost(nrm, nar, m % sv);
  for (int cap = 3; cap <= 12; ++cap) {
    int idx = 0, cum = 0;
    for (int g = 0; g < nslots; ++g) {
      d.s_lo[g] = (short)idx;
      // (where the staging has fallen six instructions behind its proportional share -- slots the epilogue filled -- one more is allowed)
      const long long due = g < s_start ? 0 : (long long)cs * (g + 1 - s_start) / (nslots - s_start - margin);
      int room = g < s_start ? 0 : cap + (due - cum >= 6 ? 1 : 0) - ecost[g];
      while (idx < ns && s_cost(nrm, nar, idx % sv) <= room &&
             (long long)(2 * cum + s_cost(nrm, nar, idx % sv)) * (nslots - s_start - margin) < (long long)2 * cs * (g + 1 - s_start)) {
        room -= s_cost(nrm, nar, idx % sv); cum += s_cost(nrm, nar, idx % sv); ++idx;
      }
    }
    if (idx == ns) break;
  }
  for (int g = nslots; g <= MAX_SLOTS; ++g) d.s_lo[g] = (short)ns;
  for (int g = 0; g < MAX_SLOTS; ++g) d.ld[g] = -1;
  if (!nar) {
    int g = 0;
    for (int k = 0; k < NITP; ++k) {
      while (d.s_lo[g + 1] <= k * sv) ++g;             // the slot of the vector's first item
      int l = g - LEAD < 0 ? 0 : g - LEAD;
      while (d.ld[l] >= 0) ++l;
      d.ld[l] = (signed char)k;
    }
  }
  for (int g = 0; g < nslots; ++g) {
    int c = ecost[g];
    for (int m = d.s_lo[g]; m < d.s_lo[g + 1]; ++m) c += s_cost(nrm, nar, m % sv);
    if (d.s_lo[g + 1] - d.s_lo[g] > d.max_s) d.max_s = (short)(d.s_lo[g + 1] - d.s_lo[g]);
    if (d.e_lo[g + 1] - d.e_lo[g] > d.max_e) d.max_e = (short)(d.e_lo[g + 1] - d.e_lo[g]);
    if (c > d.max_cost) d.max_cost = (short)c;
  }
  return d;
}
template <bool NRM, int KIND, int CI> struct DealOf { static constexpr Deal v = make_deal(NRM, KIND, CI); };
constexpr int MAXS = 8, MAXE = 4;
// the K loops are written as compile-time loops: f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>), every slot number a
// constant expression that indexes the deal table (as `#pragma unroll` loops over table look-ups the file took 20+ minutes to compile)
template <class F, int... I> __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }
static_assert(make_deal(true, 0).s_lo[288] == 13 * 21 && make_deal(false, 1).s_lo[288] == 13 * 5 && make_deal(true, 3).s_lo[72] == 13 * 21 && make_deal(true, 1, 64).s_lo[144] == 7 * 21, "deal");
static_assert(make_deal(true, 0).max_s <= MAXS && make_deal(true, 1).max_s <= MAXS && make_deal(false, 0).max_s <= MAXS && make_deal(true, 3).max_s <= MAXS && make_deal(true, 1, 64).max_s <= MAXS, "deal");
static_assert(make_deal(true, 0).max_e <= MAXE && make_deal(true, 1).max_e <= MAXE && make_deal(false, 2).max_e <= MAXE && make_deal(true, 1, 64).max_e <= MAXE + 2, "deal");
static_assert(make_deal(true, 0).max_cost <= 4 && make_deal(false, 0).max_cost <= 3 && make_deal(true, 1).max_cost <= 6, "deal: no slot above four (statistics form: six) extra instructions");
}  // namespace wst

// NRM: the input carries a pending norm + ReLU (forward); KIND: 0 no statistics, 1 forward statistics, 2 norm-backward sums (data gradient),
// 3 the NARROW form: Cout <= 32 (the heads' last conv, 128 -> J, model.py:64 / :113), fp32 NCHW output -- see the block behind the prologue
template <bool NRM, int KIND, int CI = 128>
__global__ __launch_bounds__(256, 1) void conv3x3_wstat_kernel(const WstatArgs a) {
  using namespace wst;
  typedef bf16_t T;
  typedef bf16x8 V;
  typedef Shape<CI> SH;
  constexpr int CIN = CI, KCH = SH::KCH, ITERS = SH::ITERS, PITCH = SH::PITCH, NITP = SH::NITP, PATCH_BYTES = SH::PATCH_BYTES, SLOTS = SH::SLOTS,
                HSLOTS = SH::HSLOTS, NSLOTS = SH::NSLOTS, SPP = SH::SPP, VPR = SH::VPR, KX = SH::KX, PXR = SH::PXR;
  constexpr int EP = 8;
  constexpr bool NAR = KIND == 3;
  static_assert(!NAR || CI == 128, "the narrow form exists for 128 input channels");
  __shared__ __attribute__((aligned(16))) char smem[2 * PATCH_BYTES];
  __shared__ float sbias[NAR ? 32 : 1];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave = 32-channel group
  // v_mfma_f32_32x32x16_bf16, A = weights (32 rows = this wave's channels), B = pixels (32 columns = one tile row): lane = column + 32 h,
  // h = the 8-channel half of the 16-channel K step.  (The 16x16x32 form leaves the vector ALU 8 issue cycles per 16-cycle MFMA, this one
  // 24 per 32: with ONE wave per SIMD the staging and epilogue arithmetic has to fit into exactly that shadow -- measured by elimination,
  // the 16x16x32 build ran its K loops at 1.36x the matrix-pipe time.)  The plain LDS image (pixel pitch 272 B, channel slot c at 16 c)
  // is conflict-free for these reads: the 32 lanes of a half read ONE slot of 32 consecutive pixels.
  const int pc = lane & 31, hh = lane >> 5;
  const int job = blockIdx.x & (a.njobs - 1), wgj = blockIdx.x >> a.job_shift;
  const ConvParams& p = a.job[job];
  const T* __restrict__ x = reinterpret_cast<const T*>(p.x);
  const int HW = p.H * p.W;
  const int tiles_x = p.W >> 5, tiles_img = tiles_x * (p.H >> 2), ntiles = p.B * tiles_img;
  // this workgroup's tiles: a contiguous range (consecutive tiles share halo rows: L2 / L1 hits), ranges of one XCD's workgroups adjacent
  int t, t_end;
  {
    const int nx = 8 >> a.job_shift;                             // workgroups of one job sit on `nx` XCD labels (blockIdx % 8: speed only)
    const int q = a.xq, r = a.xr, xl = wgj & (nx - 1);
    const int c = (xl < r ? xl * (q + 1) : r * (q + 1) + (xl - r) * q) + (wgj >> a.nx_shift);
    t = c * a.tiles_q + (c < a.tiles_r ? c : a.tiles_r);          // (no 64-bit division in the prologue)
    t_end = t + a.tiles_q + (c < a.tiles_r ? 1 : 0);
  }
  if (t >= t_end) return;
  // (debug build: s_memtime stamps of thread 0, 32 x int64 per workgroup: 0 start, 1 first patch staged, then per tile: half A done, half B
  // done, barrier passed; 31: end)
  auto stamp = [&](int slot) __attribute__((always_inline)) {
#ifdef PWR_DEBUG_BUILD
    if (p.stamps && tid == 0 && slot < 29) p.stamps[(size_t)blockIdx.x * 32 + slot] = (long long)__builtin_amdgcn_s_memtime();
    if (p.stamps && tid == 0 && (slot == 0 || slot == 31)) {     // start / end also by the constant 100 MHz clock: the shader clock the chip held
      p.stamps[(size_t)blockIdx.x * 32 + (slot ? 30 : 29)] = (long long)__builtin_amdgcn_s_memrealtime();
      if (slot) p.stamps[(size_t)blockIdx.x * 32 + 31] = (long long)__builtin_amdgcn_s_memtime();
    }
#endif
  };
  stamp(0);

  // ---- staging of a patch.  Thread -> 16-byte channel slot st_slot of 13 patch pixels: vectors k = 0 .. 11 are patch row k / 2, columns
  // 16 (k % 2) + st_pl (the 32 left columns of the 6 rows); vector 12 is the two right columns: row st_pl / 2, column 32 + st_pl % 2
  // (st_pl < 12; the other threads write 4 surplus pixels behind the patch).  So a vector's global address is a SCALAR row base plus a
  // per-thread column offset, its LDS address a per-thread base plus a constant, its validity a scalar row test and a per-thread column
  // test: next to no vector arithmetic per load (the norm arithmetic is what has to ride beside the MFMAs).
  const int H = p.H, W = p.W;
  const int st_pl = tid / SPP, st_slot = tid % SPP;
  const int r12 = st_pl >> 1, c12 = 32 + (st_pl & 1);                      // vector 12's patch pixel
  const int lds_st = (st_pl * PITCH) + st_slot * 16;                  // + (row * PW + PXR (k % VPR)) * PITCH
  const int lds_st12 = (st_pl < 12 ? (r12 * PW + c12) : (PP + st_pl - 12)) * PITCH + st_slot * 16;
  V sv[NITP];
  float mu[EP], sc[EP], be[EP];
  struct TileCo { int b, y0, x0; };
  // per tile, two records: where its vectors come from (LoadCo: scalar row pointers, per-thread column byte offsets) and which of them lie
  // outside the image (MaskCo).  A keep value is all-ones or zero -- in the norm form +inf or zero, as v_med3_f32(x, 0, keep) is then the
  // ReLU and the zero padding in ONE instruction (x < 0 -> 0, NaN -> 0 like v_max_f32, keep = 0 -> 0).
  constexpr unsigned KEEP = NRM ? 0x7f800000u : ~0u;
  // (addresses: raw buffer loads / stores -- a scalar 32-bit row offset in the instruction's soffset, a per-thread 32-bit byte offset, the
  // tensor's base in a resource descriptor.  As 64-bit pointers the six row bases of a tile cost ~50 scalar instructions -- 8 issue
  // cycles each with one wave per SIMD -- between the barrier and the tile's first MFMA, and every epilogue store six more.  The tensors of
  // this kernel are below 4 GiB: conv_wstat_applicable.)
  struct LoadCo {
    unsigned row[6];               // (scalar) byte offset of the clamped row y0 + r - 1 of sample b
    unsigned voff0, voff1, voff12; // byte offset of this thread's vector inside a row, for even k / odd k; k = 12: its full offset (clamped)
  };
  struct MaskCo {
    unsigned rowm[6];              // (scalar) KEEP if row y0 + r - 1 is inside the image
    unsigned keep0, keep12;        // even k: all-ones or zero by column (odd k: always inside); k = 12: KEEP or zero
  };
  const unsigned RS = (unsigned)W * (CIN * 2);                    // bytes of an image row
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(x), 0, -1, 0x00020000);
  auto load_co = [&](const TileCo& c) __attribute__((always_inline)) {
    LoadCo q;
    // rows y0 .. y0 + 3 are inside the image; only the halo rows r = 0 and r = 5 clamp
    q.row[1] = (unsigned)(c.b * H + c.y0) * RS;
    q.row[2] = q.row[1] + RS; q.row[3] = q.row[2] + RS; q.row[4] = q.row[3] + RS;
    q.row[0] = q.row[1] - (c.y0 > 0 ? RS : 0u);
    q.row[5] = q.row[4] + (c.y0 + 4 < H ? RS : 0u);
    q.voff0 = (unsigned)(max(c.x0 + st_pl - 1, 0) * (CIN * 2) + st_slot * 16);          // even k: columns x0 - 1 ... x0 + 14
    q.voff1 = VPR == 2 ? (unsigned)((c.x0 + 15 + st_pl) * (CIN * 2) + st_slot * 16) : 0u;   // (VPR = 2) odd k: columns x0 + 15 ... x0 + 30, always inside
    const int ix12 = c.x0 + c12 - 1, iy12 = c.y0 + r12 - 1;
    q.voff12 = (unsigned)(c.b * H + min(max(iy12, 0), H - 1)) * RS + (unsigned)(min(ix12, W - 1) * (CIN * 2) + st_slot * 16);
    return q;
  };
  auto mask_co = [&](const TileCo& c) __attribute__((always_inline)) {
    MaskCo q;
#pragma unroll
    for (int r = 0; r < 6; ++r) { const int iy = c.y0 + r - 1; q.rowm[r] = (iy >= 0 && iy < H) ? KEEP : 0u; }
    q.keep0 = c.x0 + st_pl - 1 >= 0 ? ~0u : 0u;
    const int ix12 = c.x0 + c12 - 1, iy12 = c.y0 + r12 - 1;
    q.keep12 = (st_pl < 12 && ix12 < W && iy12 >= 0 && iy12 < H) ? KEEP : 0u;
    return q;
  };
  auto stage_load = [&](const int k, const LoadCo& q) __attribute__((always_inline)) {
    if (k < KX) sv[k] = __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)((VPR == 2 && (k & 1)) ? q.voff1 : q.voff0), (int)q.row[k / VPR], 0));
    else sv[k] = __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)q.voff12, 0, 0));
  };
  auto stage_norm_load = [&](const TileCo& c) __attribute__((always_inline)) {
    if constexpr (NRM) {
      const size_t plane = (size_t)p.B * CIN;
      const float* st = p.in_norm + (size_t)c.b * CIN + st_slot * EP;
#pragma unroll
      for (int e = 0; e < EP; ++e) { mu[e] = st[e]; sc[e] = st[2 * plane + e]; be[e] = st[3 * plane + e]; }
    }
  };
  // A vector's way into LDS as a sequence of ITEMS (wst::s_cost), handed out behind the MFMAs by the deal table: with one wave per SIMD the
  // vector work only costs nothing while it fits into the issue slots the matrix pipe leaves, and left to itself the scheduler emits a
  // vector's ~40 staging instructions as one lump (the pipe drains for ~200 cycles, 13 x per tile).  Same arithmetic and rounding as
  // conv_patch.hip's stage_patch: fmaf(x - mu, sc, be), max 0, one rounding to bf16 (the masked lanes: zero either way).
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  constexpr int SV = sv_items(NRM);
  u32x4 so;
  float t0 = 0.f, t1 = 0.f;
  unsigned km = 0;               // the keep value of the vector in flight
  auto keep_of = [&](const int k, const MaskCo& q) __attribute__((always_inline)) {
    // (an arithmetic mask, not a select that hipcc turns into an exec-masked branch -- that would cut the K loop's scheduling region in
    // two; row validity is a scalar)
    if (k < KX) return (VPR == 2 && (k & 1)) ? q.rowm[k / VPR] : (q.keep0 & q.rowm[k / VPR]);
    return q.keep12;
  };
  // item i of vector k.  The two vector instructions of an item never depend on each other (a dependent instruction issued right behind
  // its producer waits four more cycles in the slot it was supposed to hide in).
  auto stage_item = [&](const int k, const int i, const MaskCo& q, char* patch) __attribute__((always_inline)) {
    if (i == SV - 1) {
      const int off = k < KX ? lds_st + ((k / VPR) * PW + PXR * (k % VPR)) * PITCH : lds_st12;
      *reinterpret_cast<u32x4*>(patch + off) = so;
      return;
    }
    if constexpr (!NRM) {
      if (i == 0) km = keep_of(k, q);
      so[i] = __builtin_bit_cast(u32x4, sv[k])[i] & km;
    } else {
      const int j = i / 5, o = i - 5 * j;
      if (o == 0) { t0 = (float)sv[k][2 * j]; t1 = (float)sv[k][2 * j + 1]; if (j == 0) km = keep_of(k, q); }
      else if (o == 1) { t0 = t0 - mu[2 * j]; t1 = t1 - mu[2 * j + 1]; }
      else if (o == 2) { t0 = fmaf(t0, sc[2 * j], be[2 * j]); t1 = fmaf(t1, sc[2 * j + 1], be[2 * j + 1]); }
      else if (o == 3) {      // (as asm: behind the pin of the previous item hipcc no longer knows the fma result is canonical and would put a
        // canonicalising v_max_f32 x, x, x in front of it)
        asm("v_med3_f32 %0, %0, 0, %1" : "+v"(t0) : "v"(km));
        asm("v_med3_f32 %0, %0, 0, %1" : "+v"(t1) : "v"(km));
      }
      else { bf16x2 pk; pk[0] = (bf16_t)t0; pk[1] = (bf16_t)t1; so[j] = __builtin_bit_cast(unsigned, pk); }
    }
  };
  // (an empty asm that "modifies" the staging temporaries: instruction selection orders side-effect-free arithmetic freely inside a block,
  // whatever the scheduling barriers say; this ties a slot's items between the barriers around it.  ONE per slot, behind its last item --
  // hipcc puts a wait state between such an asm and a vector instruction that reads what it "wrote")
  auto stage_pin = [&]() __attribute__((always_inline)) { asm volatile("" : "+v"(t0), "+v"(t1), "+v"(so), "+v"(km)); };
  // tile coordinates walk incrementally (no division in the loop)
  auto tile_next = [&](TileCo c) {
    c.x0 += 32;
    if (c.x0 == W) { c.x0 = 0; c.y0 += 4; if (c.y0 == H) { c.y0 = 0; ++c.b; } }
    return c;
  };

  // ---- prologue: the first patch's loads go out FIRST, the 72 weight loads behind them; the patch is normalised and written while the
  // weights are still landing (vmcnt counts in order: its wait does not cover them), and only then does anything wait for the weights.
  TileCo cur;
  {
    auto div = [](int n, int d, unsigned magic) __attribute__((always_inline)) {
      int q = (int)__umulhi((unsigned)n, magic);
      q -= q * d > n ? 1 : 0;                  // (the rounded-up reciprocal overshoots by one at most; d = 1 -- magic 2^32 - 1 -- undershoots by one)
      return q + ((q + 1) * d <= n ? 1 : 0);
    };
    cur.b = div(t, tiles_img, a.magic_img);
    const int tr = t - cur.b * tiles_img, tyi = div(tr, tiles_x, a.magic_x);
    cur.y0 = tyi * 4; cur.x0 = (tr - tyi * tiles_x) * 32;
  }
  TileCo nx1 = (t + 1 < t_end) ? tile_next(cur) : cur;            // (the last tiles re-stage themselves: branch-free K loops)
  {
    const LoadCo l0 = load_co(cur);
    stage_norm_load(cur);
#pragma unroll
    for (int k = 0; k < NITP; ++k) stage_load(k, l0);
  }
  __builtin_amdgcn_sched_barrier(0);
  stamp(25);
  // (the bias goes out BEFORE the weights: vmcnt counts in order, a wait for anything issued behind the weights is a wait for all of them)
  const int n = 32 * wn + 16 * hh;
  float bias_r[16];            // (a data gradient has no bias: KIND 2 spends these sixteen registers on the norm state instead)
  if constexpr (KIND != 2) {
    const float* bp = p.bias ? p.bias + n : reinterpret_cast<const float*>(p.w);
#pragma unroll
    for (int e = 0; e < 16; ++e) bias_r[e] = bp[e];
  } else {
#pragma unroll
    for (int e = 0; e < 16; ++e) bias_r[e] = 0.f;
  }
  __builtin_amdgcn_sched_barrier(0);
  // ---- the weights of this wave: A fragments of all 72 K steps of 16 channels (lane = row + 32 h: weight row `row`, channels 8 h .. 8 h + 7
  // of the step).  MFMA row -> output channel: the accumulator of lane (col, h) holds rows (r % 4) + 8 (r / 4) + 4 h, r = 0 .. 15; with
  // channel = 16 (row / 4 % 2) + 4 (row / 8) + row % 4 those are the 16 CONSECUTIVE channels 16 h .. 16 h + 15 of the wave's 32: two
  // 16-byte NHWC stores per pixel straight from registers.
  // Only taps 0 .. 2 are requested here.  vmcnt counts in order, so the first wait for anything issued behind the weights is a wait for ALL
  // weights requested so far: with all 288 KiB per CU requested up front, a workgroup sat ~5 k cycles between its first patch and its first
  // MFMA while they streamed in from L2.  Taps 3 .. 8 are requested from inside the first tile's first K loop, three taps (48 MFMA slots)
  // ahead of their use (load_tap below): the first tile computes while its weights arrive.
  V wreg[2 * ITERS];
  unsigned wvoff;                                                // per-lane byte offset of the lane's 16 bytes inside a fragment
  int qs0, qs1;                                                  // byte strides of q / 2 and q % 2
  {
    const int row = lane & 31;
    const int ch = NAR ? row : 32 * wn + 16 * ((row >> 2) & 1) + 4 * (row >> 3) + (row & 3);    // (narrow: every wave holds the same 32 rows)
    // (fragment-order pack, conv_mfma.hip PackDesc::order 1: fragment q of wave wn is the contiguous KiB [q / 2][wn][q % 2][lane]; the standard
    // pack serves too -- 32 pieces of 32 B per wave instruction, a ~10 000-cycle prologue)
    wvoff = p.w_frag ? (unsigned)(wn * 128 + lane) * 16u : (unsigned)(ch * 32 + hh * 8) * 2u;
    qs0 = p.w_frag ? 512 * 16 : p.CoutPad * 64; qs1 = p.w_frag ? 64 * 16 : 32;
  }
  constexpr int TAPS_AHEAD = WST_TAPS_AHEAD(NRM, KIND);
  // (`z`: an opaque zero.  With addresses the compiler can prove loop-invariant it computes all 48 in front of the tile loop and keeps them
  // in 96 registers that do not exist)
  auto load_tap = [&](const int tap, const int z) __attribute__((always_inline)) {
    const char* wb = reinterpret_cast<const char*>(p.w) + z;
#pragma unroll
    for (int q = 2 * KCH * tap; q < 2 * KCH * (tap + 1); ++q) wreg[q] = *reinterpret_cast<const V*>(wb + ((q >> 1) * qs0 + (q & 1) * qs1) + wvoff);
  };
#pragma unroll
  for (int tap = 0; tap < TAPS_AHEAD; ++tap) load_tap(tap, 0);
  if (TAPS_AHEAD < 9) load_tap(8, 0);      // (tap 8 lives in VGPRs; requested from inside the loop it cost the statistics form 27 spilled registers)
  __builtin_amdgcn_sched_barrier(0);
  {
    // the first patch into buffer 0; narrow form: each vector's registers take the SECOND tile's vector the moment they are free (its first
    // use is in slot 0 of the first K loop: the vectors requested first get the rest of this block, ~2 000 cycles, to arrive)
    const MaskCo m0 = mask_co(cur);
    const LoadCo l1 = load_co(nx1);
#pragma unroll
    for (int k = 0; k < NITP; ++k) {
#pragma unroll
      for (int i = 0; i < SV; ++i) stage_item(k, i, m0, smem);
      stage_pin();
      if (NAR) stage_load(k, l1);
    }
  }
  if constexpr (NAR) {
    if (tid < 32) sbias[tid] = (p.bias && tid < p.Cout) ? p.bias[tid] : 0.f;
  }
  if constexpr (KIND != 2) {      // (no bias: zeros.  Here, behind the staging arithmetic, so that the wait for the loaded values stands here too)
#pragma unroll
    for (int e = 0; e < 16; ++e) { asm volatile("" : "+v"(bias_r[e])); bias_r[e] = p.bias ? bias_r[e] : 0.f; }
  }
  stamp(26);
  __syncthreads();
  stamp(27);
  // Register files: the vector ALU only reaches the 256 architectural VGPRs, the MFMA reads its operands from either file.  Left alone the
  // allocator puts most weights into VGPRs, runs out of them where the staging arithmetic lives and shuffles weights through AGPR spill
  // slots (4 v_accvgpr_mov per MFMA operand, in the middle of the K loop).  So: the weights of taps 0 .. 7 are DEFINED in AGPRs (all 256
  // of them: an empty asm with a "+a" operand) and stay there; tap 8, the ACCUMULATORS (the file is built with -amdgpu-mfma-vgpr-form: the
  // epilogue's vector instructions read them directly, no v_accvgpr_read per value) and everything else the vector ALU touches share the VGPRs.
  // WHERE those empty asms stand is where a wave first waits for the weights: in front of each tap's first MFMA of a tile's first K loop,
  // not in front of the loop -- the first tile starts on tap 0 while taps 1 .. 8 are still streaming in from L2 (288 KiB per CU; waiting
  // for all of it before the first MFMA was ~5 k of a workgroup's ~60 k cycles).  In later tiles the waits they imply are already satisfied.
  auto pin_tap = [&](const int tap) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 2 * KCH * tap; q < 2 * KCH * (tap + 1); ++q)
      if (q < 2 * WST_AGPR_STEPS) asm volatile("" : "+a"(wreg[q]));
  };
  stamp(1);
  int tile_no = 0;

  constexpr int RING = WST_RING;                                 // B fragments: a ring of RING (read RING slots = 32 RING matrix-pipe cycles ahead of their use)
  V pf[RING];

  if constexpr (NAR) {
    // ---- the NARROW form: all four waves hold the SAME 32 output channels (Cout <= 32, pack padded to 32 rows; MFMA row = channel) and
    // split the tile's four rows: wave w accumulates tile row w, 72 slots per tile, one accumulator block.  conv_patch.hip ran this shape
    // (the heads' 128 -> J conv) with a barrier per 32-channel K step around TWO MFMAs per wave: 29.6 us for a quarter of the dominant
    // conv's FLOPs.  Here the K loop has no barrier and the staging of the next patch is what bounds it (3.8 micro-ops per MFMA), so the
    // staging pipeline is TWO tiles deep: the global load of vector k of tile n + 2 is issued into the registers of vector k of tile n + 1
    // the moment that vector's last micro-op has written it to LDS -- a whole tile (~2.5 k cycles) ahead of its use, no second register set.
    // Same K order and operands as conv_patch.hip's <4, 1, 1, 1> form (tap-major, two 16-channel halves per K step) => bit-identical output.
    const char* fb = smem + pc * PITCH + hh * 16 + wn * (PW * PITCH);
    auto frag_load_n = [&](const int g, const char* base) __attribute__((always_inline)) {
      const int it = g >> 1, ss = g & 1, tap = it / KCH, kch = it - tap * KCH, ky = tap / 3, kx = tap - ky * 3;
      pf[g % RING] = *reinterpret_cast<const V*>(base + (ky * PW + kx) * PITCH + (4 * kch + 2 * ss) * 16);
    };
    // accumulator register r of lane (col, h) = channel (r % 4) + 8 (r / 4) + 4 h of pixel (row wn, col); the bias waits in LDS (sixteen
    // vector registers for it were sixteen too many in the prologue, beside 288 weights and 13 vectors in flight)
    f32x16 accn;
    int buf = 0;
    for (; t < t_end; ++t) {
      const TileCo nx2 = (t + 2 < t_end) ? tile_next(nx1) : nx1;
      const LoadCo l2 = load_co(nx2);
      const MaskCo m1_ = mask_co(nx1);
      const char* pb = fb + buf * PATCH_BYTES;
      char* nb = smem + (buf ^ 1) * PATCH_BYTES;
#pragma unroll
      for (int g = 0; g < RING; ++g) frag_load_n(g, pb);
      if (nx1.b != cur.b) stage_norm_load(nx1);                    // (its first use is in slot 0: only when the sample changes)
      __builtin_amdgcn_sched_barrier(0);
      static_for<NSLOTS>([&](auto SL) __attribute__((always_inline)) {
        constexpr int sl = decltype(SL)::value;
        constexpr int m0 = DealOf<NRM, KIND, CI>::v.s_lo[sl], m1 = DealOf<NRM, KIND, CI>::v.s_lo[sl + 1];
        if constexpr (sl % (2 * KCH) == 0) {
          if (sl / (2 * KCH) + TAPS_AHEAD < 8 && tile_no == 0) { int z = tile_no; asm volatile("" : "+s"(z)); load_tap(sl / (2 * KCH) + TAPS_AHEAD, z); }
          pin_tap(sl / (2 * KCH));
        }
        if (WST_DBG & 8) asm volatile("" : "+v"(pf[sl % RING]));
        else if (sl == 0) accn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[sl], pf[sl % RING], f32x16{}, 0, 0, 0);
        else accn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[sl], pf[sl % RING], accn, 0, 0, 0);
        if (sl + RING < NSLOTS && !(WST_DBG & 4)) frag_load_n(sl + RING, pb);
        if constexpr (m1 > m0 && !(WST_DBG & 1)) {
          static_for<m1 - m0>([&](auto J) __attribute__((always_inline)) {
            constexpr int m = m0 + decltype(J)::value, k = m / SV, i = m - k * SV;
            stage_item(k, i, m1_, nb);
            if (i == SV - 1) stage_load(k, l2);
          });
          stage_pin();
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      stamp(2 + 3 * tile_no);
      {
        float* yo = p.y_nchw + ((size_t)cur.b * p.Cout * H + (size_t)(cur.y0 + wn)) * W + cur.x0 + pc;
        float ov[16];             // (all sixteen bias reads in flight at once; read inside the predicated stores they were sixteen LDS round trips)
#pragma unroll
        for (int r = 0; r < 16; ++r) ov[r] = accn[r] + sbias[(r & 3) + 8 * (r >> 2) + 4 * hh];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int chn = (r & 3) + 8 * (r >> 2) + 4 * hh;
          if (chn < p.Cout) yo[(size_t)chn * HW] = ov[r];
        }
      }
      stamp(3 + 3 * tile_no);
      __syncthreads();      // the next patch is complete and visible; every wave has left this tile's K loop
      stamp(4 + 3 * tile_no); ++tile_no;
      cur = nx1; nx1 = nx2;
      buf ^= 1;
    }
    stamp(31);
    return;
  }

  f32x16 acc[4];                                                 // one per tile row (32 pixels x this wave's 32 channels)
  const char* fbase = smem + pc * PITCH + hh * 16;               // per-lane part of every B-fragment address (+ buffer, tap, K step: constants)
  // A tile is computed as two HALVES (tile rows 0 - 1, then 2 - 3), each a K loop of 36 steps x 4 MFMAs: while a half accumulates, the
  // EPILOGUE of the half finished before it (the other 32 accumulator registers) rides in the MFMAs' issue shadow next to the staging --
  // during half A of tile n the rows 2 - 3 of tile n - 1, during half B the rows 0 - 1 of tile n.  Nothing of the epilogue is exposed but the
  // last half of a workgroup's last tile.
  // A SLOT is one MFMA (32 matrix-pipe cycles): slot sl = 4 * step + 2 * ss + blk of a half multiplies the 16-channel half ss of K step `step`
  // into tile row 2 * half + blk.  Fragments: a ring of eight; the fragment of slot g + 8 is read into the registers of slot g's right
  // behind the MFMA that consumed them (256 matrix-pipe cycles ahead of its own use).
  auto frag_load1 = [&](const int g, const char* base) __attribute__((always_inline)) {      // g = 144 * half + slot
    const int half = g / (ITERS * 4), sl = g - half * ITERS * 4, it = sl >> 2, ss = (sl >> 1) & 1, row = 2 * half + (sl & 1);
    const int tap = it / KCH, kch = it - tap * KCH;
    const int ky = tap / 3, kx = tap - ky * 3;
    pf[g % RING] = *reinterpret_cast<const V*>(base + ((row + ky) * PW + kx) * PITCH + (4 * kch + 2 * ss) * 16);
  };

  // ---- epilogue: lane (col, h) holds channels n .. n + 15 of the pixels (row, col) of the tile: bias, one rounding, two 16-byte NHWC stores
  // per row, as micro-ops of two independent vector instructions (E_ROW per tile row).  A store's address is a scalar base (tile, row) plus a
  // per-lane constant offset.
  const unsigned yoff = (unsigned)(pc * COUT + n) * 2u;
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rnb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.nb_y), 0, -1, 0x00020000);      // (KIND 2)
  // KIND 1 (forward statistics of the norm that follows, conv_common.h EpiStats): per 8-channel slot and pixel column li = col % 16 the old
  // kernel's thread adds the pixels (it, li), it = 0 .. 7 (= tile row it / 2, column 16 (it % 2) + li), IN THAT ORDER, then a butterfly over
  // li.  Here lane (col, h) holds 16 channels of the pixels (row, col): the lanes col and col ^ 16 (rows 16 lanes apart in the wave) exchange
  // half of what they hold -- ONE v_permlane16_swap per packed dword: the lane with col < 16 keeps its channels 0 - 7 and receives its
  // partner's, the other one keeps 8 - 15 -- and every lane then owns ONE slot (channels n + 8 (col / 16) ...) of the pixels it = 2 row,
  // 2 row + 1 of column li: exactly the old thread's set, consumed in the old order; the final butterfly runs over the 16 lanes of a row.
  // Same sums, bit for bit (tests/test_00_kat_gpu.py holds both kernels to one digest).
  // KIND 2 (this launch is a data gradient g; sums of the norm backward of the tensor y it belongs to: relu-masked g and g * xhat): the same
  // exchange and order; the forward activations y of the lane's slot and two pixels come straight from global memory (two 16-byte loads per
  // tile row, issued with the row's first micro-op), the norm state of the sample sits in 32 registers, re-read at the half boundary.
  // (wst::e_st: per tile row 2 swaps + 2 pixels x 4 channel pairs x (4 | 9) items; e_a0: tile row 0, the shift (the tile's first pixel) to every
  // lane of the slot; e_bf: the butterfly over the 16 lanes of a row, 6 steps x 8 pairs of sums; half A = rows 2 - 3 of the previous tile + its
  // butterfly, half B = rows 0 - 1)
  constexpr int E_A0 = e_a0(KIND, CI), E_BF = e_bf(KIND, CI), E_ROW = e_row(KIND);
  float f0 = 0.f, f1 = 0.f, f2 = 0.f, f3 = 0.f;
  u32x4 eo, oL, oH;
  float s1[8], s2[8], a0[8];
  bool gc0 = false, gc1 = false;
  float a1[8], a2[8], a3[8], g0 = 0.f, g1 = 0.f;                  // (KIND 2: rstd, scale, beta of the slot's channels; the masked gradients of a pair)
  // (KIND 2: y of the slot's channels at a row's two pixels, two sets: a row's loads are issued with the FIRST micro-op of the row before it --
  // some 45 slots = 1400 cycles ahead; issued with its own row they arrived ~1000 cycles late, 4 x per tile: the tensor comes from HBM)
  u32x4 yvA0, yvA1, yvB0, yvB1;
  const bool nb_relu = p.nb_relu != 0;
  const int slot_ch = n + 8 * ((lane >> 4) & 1);                  // first channel of the statistics slot this lane owns after the exchange
  auto pk2 = [](float lo, float hi) __attribute__((always_inline)) { bf16x2 v; v[0] = (bf16_t)lo; v[1] = (bf16_t)hi; return __builtin_bit_cast(unsigned, v); };
  auto lo_f = [](unsigned u) __attribute__((always_inline)) { return __builtin_bit_cast(float, u << 16); };
  auto hi_f = [](unsigned u) __attribute__((always_inline)) { return __builtin_bit_cast(float, u & 0xffff0000u); };
  // accumulator register r of a lane = channel n + r (the row -> channel assignment above).  No dependent pair inside a micro-op.
  auto epi_micro = [&](const int row, const int u, const TileCo& c, const TileCo& cn) __attribute__((always_inline)) {
    if (u < 14) {
      const int e8 = u / 7, v = u - 7 * e8, c0 = 8 * e8;          // the 8 channels c0 .. c0 + 7: seven micro-ops
      if (v == 0) { f0 = acc[row][c0 + 0] + bias_r[c0 + 0]; f1 = acc[row][c0 + 1] + bias_r[c0 + 1]; }
      else if (v == 1) { f2 = acc[row][c0 + 2] + bias_r[c0 + 2]; f3 = acc[row][c0 + 3] + bias_r[c0 + 3]; }
      else if (v == 2) { eo[0] = pk2(f0, f1); eo[1] = pk2(f2, f3); }
      else if (v == 3) { f0 = acc[row][c0 + 4] + bias_r[c0 + 4]; f1 = acc[row][c0 + 5] + bias_r[c0 + 5]; }
      else if (v == 4) { f2 = acc[row][c0 + 6] + bias_r[c0 + 6]; f3 = acc[row][c0 + 7] + bias_r[c0 + 7]; }
      else if (v == 5) { eo[2] = pk2(f0, f1); eo[3] = pk2(f2, f3); }
      else {
        __builtin_amdgcn_raw_buffer_store_b128(eo, ry, (int)(yoff + e8 * 16), (int)((unsigned)((c.b * H + c.y0 + row) * W + c.x0) * (COUT * 2)), 0);
        if (KIND != 0) { if (e8 == 0) oL = eo; else oH = eo; }
      }
      if (KIND == 2 && u == 0) {       // y of the NEXT row in processing order (rows 0, 1 of `cn` = the current tile; 2, 3 of the tile c)
        const int nr = (row + 1) & 3;
        const TileCo& cc = row == 3 ? cn : c;
        // (16: always the tensor's first row -- cache-hot: what the loads' latency costs.  Measured: nothing)
        const int yso = (WST_DBG & 16) ? 0 : (int)((unsigned)((cc.b * H + cc.y0 + nr) * W + cc.x0) * (COUT * 2));
        const int o16 = (int)((unsigned)((lane & 15) * COUT + slot_ch) * 2u);
        if (nr & 1) { yvB0 = __builtin_amdgcn_raw_buffer_load_b128(rnb, o16, yso, 0); yvB1 = __builtin_amdgcn_raw_buffer_load_b128(rnb, o16 + 16 * COUT * 2, yso, 0); }
        else { yvA0 = __builtin_amdgcn_raw_buffer_load_b128(rnb, o16, yso, 0); yvA1 = __builtin_amdgcn_raw_buffer_load_b128(rnb, o16 + 16 * COUT * 2, yso, 0); }
      }
      return;
    }
    if constexpr (KIND == 2) {
      const int w = u - 14;
      if (w < 2) {
#pragma unroll
        for (int j = 2 * w; j < 2 * w + 2; ++j) {
          const auto r = __builtin_amdgcn_permlane16_swap(oL[j], oH[j], false, false);
          oL[j] = r[0]; oH[j] = r[1];
        }
        return;
      }
      const int v = w - 2;                                         // 0 .. 71: pixel (L', then H') x channel pair x 9 micro-ops
      const int x = v / 36, j = (v / 9) & 3, o = v % 9;
      const unsigned gp = x ? oH[j] : oL[j], yp = (row & 1) ? (x ? yvB1[j] : yvB0[j]) : (x ? yvA1[j] : yvA0[j]);
      // EpiStats::add_pre: gg = g unless relu && !(fma(y - mean, scale, beta) > 0); s1 += gg; s2 = fma(gg, (y - mean) * rstd, s2)
      if (o == 0) { f0 = lo_f(yp); f1 = hi_f(yp); }
      else if (o == 1) { g0 = lo_f(gp); g1 = hi_f(gp); }
      else if (o == 2) { f0 = f0 - a0[2 * j]; f1 = f1 - a0[2 * j + 1]; }
      else if (o == 3) { f2 = fmaf(f0, a2[2 * j], a3[2 * j]); f3 = fmaf(f1, a2[2 * j + 1], a3[2 * j + 1]); }
      // (the two compares in one item, the two selects in the next: a select right behind the compare that wrote its mask register waits
      // two states for it.  No ReLU: nb_state_load made f2, f3 = 1)
      else if (o == 4) { gc0 = f2 > 0.f; gc1 = f3 > 0.f; }
      else if (o == 5) { g0 = gc0 ? g0 : 0.f; g1 = gc1 ? g1 : 0.f; }
      else if (o == 6) { s1[2 * j] += g0; s1[2 * j + 1] += g1; }
      else if (o == 7) { f0 = f0 * a1[2 * j]; f1 = f1 * a1[2 * j + 1]; }
      else { s2[2 * j] = fmaf(g0, f0, s2[2 * j]); s2[2 * j + 1] = fmaf(g1, f1, s2[2 * j + 1]); }
      return;
    }
    if constexpr (KIND == 1) {
      const int w = u - 14;
      if (w < 2) {                       // the exchange: dwords 2 w, 2 w + 1
#pragma unroll
        for (int j = 2 * w; j < 2 * w + 2; ++j) {
          const auto r = __builtin_amdgcn_permlane16_swap(oL[j], oH[j], false, false);
          oL[j] = r[0]; oH[j] = r[1];
        }
        return;
      }
      if (w < 2 + E_A0 && row == 0) {    // (tile row 0 only) the shift = the slot's values at the tile's first pixel: from lane li = 0 of this row
        const int z = w - 2;
        if constexpr (CI == 64) {
          // the fp32 value (accumulator + bias) of channel z of the lane's slot at pixel (0, 0): lane col = 0 of the same K half holds the
          // 16 channels n .. n + 15 of that pixel; the slot of the lanes 16 .. 31 / 48 .. 63 is its upper eight
          const float lo = __shfl(acc[0][z] + bias_r[z], lane & 32, 64), hi = __shfl(acc[0][8 + z] + bias_r[8 + z], lane & 32, 64);
          a0[z] = (lane & 16) ? hi : lo;
        } else {
          if (z < 2) { eo[2 * z] = __shfl(oL[2 * z], lane & 48, 64); eo[2 * z + 1] = __shfl(oL[2 * z + 1], lane & 48, 64); }
          else { const int j = z - 2; a0[2 * j] = lo_f(eo[j]); a0[2 * j + 1] = hi_f(eo[j]); }
        }
        return;
      }
      // 0 .. 31: channel pair j = v / 8; inside it the four micro-ops {unpack, - shift, s1, s2} of the row's two pixels ALTERNATE (x = v % 2),
      // each with its own temporaries: consecutive micro-ops never depend on each other; per sum the pixel order is kept
      const int v = w - 2 - (row == 0 ? E_A0 : 0);
      if (v < 0 || v >= 32) return;
      const int j = v >> 3, o = (v & 7) >> 1, x = v & 1;
      const unsigned pkd = x ? oH[j] : oL[j];
      float& q0 = x ? f2 : f0; float& q1 = x ? f3 : f1;
      if (o == 0) { q0 = lo_f(pkd); q1 = hi_f(pkd); }
      else if (o == 1) { q0 = q0 - a0[2 * j]; q1 = q1 - a0[2 * j + 1]; }
      else if (o == 2) { s1[2 * j] += q0; s1[2 * j + 1] += q1; }
      else { s2[2 * j] = fmaf(q0, q0, s2[2 * j]); s2[2 * j + 1] = fmaf(q1, q1, s2[2 * j + 1]); }
    }
  };
  // The butterfly over the 16 lanes of a row (xor 1, 2, 4, 8 like the old kernel's __shfl_xor loop, which hipcc turns into ds_bpermute_b32:
  // 64 LDS round trips, each behind a wait that also drains the fragment reads) as DPP moves: xor 1 / 2 = quad_perm, xor 8 = row_ror:8,
  // xor 4 = row_shl:4 into banks 0 and 2 + row_shr:4 into banks 1 and 3 (two moves and an add).  A micro-op handles one pair of sums
  // 2 j, 2 j + 1 of (s1[0 .. 7], s2[0 .. 7]).
  float bt0_ = 0.f, bt1_ = 0.f, bt2 = 0.f, bt3 = 0.f;
  // As inline asm: through __builtin_amdgcn_update_dpp hipcc emitted v_mov_b32 (a copy), s_nop, v_mov_b32_dpp, v_add_f32 for what ONE
  // v_add_f32_dpp does (the DPP operand on the add itself) -- 13 instructions per sum where 6 do, and every instruction of this loop costs
  // issue time.  The hazard the compiler's s_nop covered -- a DPP operand read needs two wait states behind a vector write of that
  // register -- is kept by the ORDER of the items: the same pair of sums comes up again eight items later (xor 1 -> xor 2 -> xor 4, xor 4
  // -> xor 8), the three xor-4 items of a pair alternate with another pair's (two instructions in between).
#define PWR_DPP_ADD(x, ctrl) asm volatile("v_add_f32_dpp %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf" : "+v"(x))
#define PWR_DPP_MOV(d, x, ctrl, bank) asm volatile("v_mov_b32_dpp %0, %1 " ctrl " row_mask:0xf bank_mask:" bank : "+v"(d) : "v"(x))
#define PWR_DPP_ADD3(d, x, y, ctrl) asm volatile("v_add_f32_dpp %0, %1, %2 " ctrl " row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(x), "v"(y))       // d = x[shifted lane] + y
  auto bfly_micro = [&](const int b) __attribute__((always_inline)) {
    if constexpr (KIND != 0) {
      // (b = 0 .. 15: xor 1, xor 2 over the eight pairs; 16 .. 39: per pair its three xor-4 ops in a row -- they share the two temporaries --;
      // 40 .. 47: xor 8)
      // (the xor-4 ops of two pairs alternate, each pair with its own two temporaries: no dependent neighbours)
      const int step = b < 16 ? (b >> 3) : (b < 40 ? 2 + ((b - 16) % 6) / 2 : 5);
      const int j = b < 16 ? (b & 7) : (b < 40 ? 2 * ((b - 16) / 6) + ((b - 16) & 1) : b - 40);
      float& x0 = j < 4 ? s1[2 * j] : s2[2 * (j - 4)];
      float& x1 = j < 4 ? s1[2 * j + 1] : s2[2 * (j - 4) + 1];
      float& bt0 = (j & 1) ? bt2 : bt0_; float& bt1 = (j & 1) ? bt3 : bt1_;
      if constexpr (CI == 64) {
        // The kernel this form replaces (conv_patch.hip's two-pass epilogue, EpiStats::finish) adds the four groups of four columns ONE AFTER
        // THE OTHER -- ((g0 + g1) + g2) + g3 -- where the one-pass epilogue's butterfly pairs them: after xor 1 and xor 2 every lane holds
        // its group's sum; lane 0 of the row collects the others with three shifted adds (only that lane's result is written).  40 items.
        if (step == 0) { PWR_DPP_ADD(x0, "quad_perm:[1,0,3,2]"); PWR_DPP_ADD(x1, "quad_perm:[1,0,3,2]"); }
        else if (step == 1) { PWR_DPP_ADD(x0, "quad_perm:[2,3,0,1]"); PWR_DPP_ADD(x1, "quad_perm:[2,3,0,1]"); }
        else if (step == 2) { PWR_DPP_ADD3(bt0, x0, x0, "row_shl:4"); PWR_DPP_ADD3(bt1, x1, x1, "row_shl:4"); }
        else if (step == 3) { PWR_DPP_ADD3(bt0, x0, bt0, "row_shl:8"); PWR_DPP_ADD3(bt1, x1, bt1, "row_shl:8"); }
        else if (step == 4) { PWR_DPP_ADD3(x0, x0, bt0, "row_shl:12"); PWR_DPP_ADD3(x1, x1, bt1, "row_shl:12"); }
        return;
      }
      if (step == 0) { PWR_DPP_ADD(x0, "quad_perm:[1,0,3,2]"); PWR_DPP_ADD(x1, "quad_perm:[1,0,3,2]"); }
      else if (step == 1) { PWR_DPP_ADD(x0, "quad_perm:[2,3,0,1]"); PWR_DPP_ADD(x1, "quad_perm:[2,3,0,1]"); }
      else if (step == 2) { PWR_DPP_MOV(bt0, x0, "row_shl:4", "0x5"); PWR_DPP_MOV(bt1, x1, "row_shl:4", "0x5"); }
      else if (step == 3) { PWR_DPP_MOV(bt0, x0, "row_shr:4", "0xa"); PWR_DPP_MOV(bt1, x1, "row_shr:4", "0xa"); }
      else if (step == 4) { x0 += bt0; x1 += bt1; }
      else { PWR_DPP_ADD(x0, "row_ror:8"); PWR_DPP_ADD(x1, "row_ror:8"); }
    }
  };
#undef PWR_DPP_ADD
#undef PWR_DPP_MOV
#undef PWR_DPP_ADD3
  // the finished sums of tile c: lane li = 0 of every row writes its slot's slab entries; then the sums restart
  auto stats_write = [&](const TileCo& c, const bool write) __attribute__((always_inline)) {
    if constexpr (KIND != 0) {
      if (write && (lane & 15) == 0) {       // (not for the garbage the first tile's stand-in "previous half" produced)
        const int tr = (c.y0 >> 2) * tiles_x + (c.x0 >> 5);
        const size_t srow = (size_t)c.b * (p.st_nchunks ? p.st_nchunks : tiles_img) + p.st_chunk0 + tr;
        float* out = (KIND == 1 ? p.st_partial + (srow * 3) * COUT : p.nb_partial + (srow * 2) * COUT) + slot_ch;
#pragma unroll
        for (int e = 0; e < 8; ++e) { out[e] = s1[e]; out[COUT + e] = s2[e]; if (KIND == 1) out[2 * COUT + e] = a0[e]; }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    }
  };
  // (KIND 2) mean, rstd, scale, beta of sample b for the slot's channels (conv_common.h: nb_state = [4][B][C])
  auto nb_state_load = [&](const int b) __attribute__((always_inline)) {
    if constexpr (KIND == 2) {
      const size_t plane = (size_t)p.B * COUT;
      const float* st = p.nb_state + (size_t)b * COUT + slot_ch;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        a0[e] = st[e]; a1[e] = st[plane + e];
        // (no ReLU on that norm: scale 0, beta 1 make the sign test of the masked gradient pass for every element -- one compare + select
        // per element instead of compare + scalar and + select)
        a2[e] = nb_relu ? st[2 * plane + e] : 0.f; a3[e] = nb_relu ? st[3 * plane + e] : 1.f;
      }
    }
  };
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; a0[e] = 0.f; }
  nb_state_load(cur.b);

  int buf = 0;
  TileCo prev = cur;            // (first tile: the "previous tile's" half epilogue stores garbage where this tile's own epilogue writes later)
  constexpr int EHA = e_half(KIND, 0, CI);
  // (the next tile's offsets and masks are computed BEFORE the barrier that ends a tile -- there a wave waits for the others anyway --,
  // not between the barrier and the tile's first MFMA; the empty asm keeps them there)
  LoadCo l1 = load_co(nx1);
  MaskCo m1_ = mask_co(nx1);
  auto pin_co = [&]() __attribute__((always_inline)) {
    asm volatile("" : "+v"(l1.voff0), "+v"(l1.voff1), "+v"(l1.voff12), "+v"(m1_.keep0), "+v"(m1_.keep12));
  };
  for (; t < t_end; ++t) {
    const char* pb = fbase + buf * PATCH_BYTES;
    char* nb = smem + (buf ^ 1) * PATCH_BYTES;
#pragma unroll
    for (int g = 0; g < RING; ++g) frag_load1(g, pb);
    if (nx1.b != cur.b) stage_norm_load(nx1);                      // (its first use is in slot 0: only when the sample changes)
    __builtin_amdgcn_sched_barrier(0);
    static_for<2>([&](auto HALF) __attribute__((always_inline)) {
      constexpr int half = decltype(HALF)::value;
      static_for<HSLOTS>([&](auto SL) __attribute__((always_inline)) {
        // a slot: the MFMA, the read that refills its fragment RING slots ahead, this slot's items of the staging and of the other half's
        // epilogue (the deal table), a full scheduling barrier: the emitted order IS this order
        constexpr int sl = decltype(SL)::value;
        constexpr int g = half * HSLOTS + sl, it = sl >> 2, ss = (sl >> 1) & 1, row = 2 * half + (sl & 1);
        constexpr int m0 = DealOf<NRM, KIND, CI>::v.s_lo[g], m1 = DealOf<NRM, KIND, CI>::v.s_lo[g + 1];
        constexpr int e0 = DealOf<NRM, KIND, CI>::v.e_lo[g], e1 = DealOf<NRM, KIND, CI>::v.e_lo[g + 1];
        if constexpr (half == 0 && sl % (4 * KCH) == 0) {
          if (sl / (4 * KCH) + TAPS_AHEAD < 8 && tile_no == 0) { int z = tile_no; asm volatile("" : "+s"(z)); load_tap(sl / (4 * KCH) + TAPS_AHEAD, z); }     // (first tile only: a uniform branch)
          pin_tap(sl / (4 * KCH));
        }
        if (WST_DBG & 8) asm volatile("" : "+v"(pf[g % RING]));
        else if (it == 0 && ss == 0) acc[row] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[2 * it + ss], pf[g % RING], f32x16{}, 0, 0, 0);
        else acc[row] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[2 * it + ss], pf[g % RING], acc[row], 0, 0, 0);
        if (g + RING < SLOTS && !(WST_DBG & 4)) frag_load1(g + RING, pb);
        if constexpr (DealOf<NRM, KIND, CI>::v.ld[g] >= 0 && !(WST_DBG & 1)) stage_load(DealOf<NRM, KIND, CI>::v.ld[g], l1);
        if constexpr (m1 > m0 && !(WST_DBG & 1)) {
          static_for<m1 - m0>([&](auto J) __attribute__((always_inline)) {
            constexpr int m = m0 + decltype(J)::value, k = m / SV, i = m - k * SV;
            stage_item(k, i, m1_, nb);
          });
          stage_pin();
        }
        if constexpr (e1 > e0 && !(WST_DBG & 2)) {
          // (half B finishes tile rows 0 - 1 of this tile -- row 0 carries the E_A0 extra ops --, half A rows 2 - 3 of the previous one and
          // then the butterfly of its sums)
          constexpr int r0n = E_ROW + (half ? E_A0 : 0);
          static_for<e1 - e0>([&](auto J) __attribute__((always_inline)) {
            constexpr int ea = e0 + decltype(J)::value, eu = half ? ea - EHA : ea;
            if (eu < r0n) epi_micro(half ? 0 : 2, eu, half ? cur : prev, cur);
            else if (eu - r0n < E_ROW) epi_micro(half ? 1 : 3, eu - r0n, half ? cur : prev, cur);
            else bfly_micro(eu - r0n - E_ROW);
          });
          // (one tie per slot, behind its last item)
          asm volatile("" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(eo));
          if (KIND != 0) {
            asm volatile("" : "+v"(oL), "+v"(oH), "+v"(bt0_), "+v"(bt1_), "+v"(bt2), "+v"(bt3));
            if (KIND == 2) asm volatile("" : "+v"(g0), "+v"(g1));
            // (the butterfly's DPP instructions are volatile asm themselves: they stay where they are written without a tie)
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      stamp(2 + 3 * tile_no + half);
      if (half == 0) {
        stats_write(prev, tile_no > 0);                            // the previous tile's rows 2 - 3 and the butterfly went in during this half A
        if (KIND == 2 && cur.b != prev.b) nb_state_load(cur.b);    // (first needed some twenty slots into half B)
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    prev = cur;
    cur = nx1; nx1 = (t + 2 < t_end) ? tile_next(nx1) : nx1;
    l1 = load_co(nx1); m1_ = mask_co(nx1);
    pin_co();
    __syncthreads();      // the next patch is complete and visible; every wave has left this tile's K loops
    stamp(4 + 3 * tile_no); ++tile_no;
    buf ^= 1;
  }
  // the last tile's second half
#pragma unroll
  for (int eu = 0; eu < 2 * E_ROW; ++eu) epi_micro(2 + eu / E_ROW, eu % E_ROW, prev, prev);
#pragma unroll
  for (int b = 0; b < E_BF; ++b) bfly_micro(b);
  stats_write(prev, true);
  stamp(31);
}

bool conv_wstat_shape(int B, int H, int W, int Cin, int Cout, int ksize, int stride, int dtype) {
  ConvParams p;
  p.x = nullptr; p.w = nullptr; p.bias = nullptr; p.in_norm = nullptr; p.residual = nullptr; p.y = (void*)1; p.y_nchw = nullptr;
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.CoutPad = Cout; p.Ho = H; p.Wo = W;
  p.ksize = ksize; p.stride = stride; p.pad = ksize / 2; p.mode = 0; p.relu_in = 1; p.KCH = Cin / 32; p.M = B * H * W;
  return conv_wstat_applicable(p, dtype);
}

bool conv_wstat_applicable(const ConvParams& p, int dtype) {
  const bool on = (PWR_DBG_ENV("PWR_WSTAT", 1) != 0) &&          // (debug build: read on every call, so one process can A/B the two kernels THROUGH THE KERNEL ENTRY POINTS with standard packs; an engine plan bakes fragment-order packs in when it is built, so the switch must be set before the plan is created -- toggling it on a live plan makes the heads' launches return PWR_EINVAL)
                  (long long)p.B * p.H * p.W * 256 < (1ll << 32);            // (32-bit byte offsets into x and y)
  const int min_tiles = PWR_DBG_ENV("PWR_WSTAT_MIN_TILES", 16);
  // (64 input channels -- the stem's 64 -> 128 conv -- from the standard pack, forward forms only)
  const bool cin_ok = p.Cin == 128 || (p.Cin == 64 && !p.w_frag && !p.nb_partial && PWR_DBG_ENV("PWR_WSTAT_C64", 1) != 0);
  return on && dtype == PWR_BF16 && p.mode == 0 && p.ksize == 3 && p.stride == 1 && p.pad == 1 && cin_ok && p.Cout == 128 &&
         p.CoutPad == 128 && p.W % 32 == 0 && p.H % 4 == 0 && p.y != nullptr && !p.y_nchw && !p.residual && (!p.in_norm || p.relu_in) &&
         !(p.st_partial && p.nb_partial) && !(p.nb_partial && (p.in_norm || p.bias || !PWR_DBG_ENV("PWR_WSTAT_NB", 0))) && p.B * (p.H / 4) * (p.W / 32) >= min_tiles;
}

// the narrow form (KIND 3): 128 -> Cout <= 32 channels, fp32 NCHW output only (the heads' last conv)
bool conv_wstat_narrow_applicable(const ConvParams& p, int dtype) {
  const bool on = (PWR_DBG_ENV("PWR_WSTAT", 1) != 0) && (PWR_DBG_ENV("PWR_WSTAT_NARROW", 1) != 0) && (long long)p.B * p.H * p.W * 256 < (1ll << 32);
  return on && dtype == PWR_BF16 && p.mode == 0 && p.ksize == 3 && p.stride == 1 && p.pad == 1 && p.Cin == 128 && p.Cout <= 32 &&
         p.CoutPad == 32 && p.W % 32 == 0 && p.H % 4 == 0 && p.y == nullptr && p.y_nchw != nullptr && !p.residual && !p.w_frag &&
         (!p.in_norm || p.relu_in) && !p.st_partial && !p.nb_partial && p.B * (p.H / 4) * (p.W / 32) >= PWR_DBG_ENV("PWR_WSTAT_MIN_TILES", 16);
}

template <bool NRM>
static void launch_kind(const WstatArgs& a, int kind, dim3 grid, hipStream_t s) {
  if (a.job[0].Cin == 64) {
    if (kind == 0) hipLaunchKernelGGL((conv3x3_wstat_kernel<NRM, 0, 64>), grid, dim3(256), 0, s, a);
    else if (kind == 1) hipLaunchKernelGGL((conv3x3_wstat_kernel<NRM, 1, 64>), grid, dim3(256), 0, s, a);
    return;
  }
  if (kind == 0) hipLaunchKernelGGL((conv3x3_wstat_kernel<NRM, 0>), grid, dim3(256), 0, s, a);
  else if (kind == 1) hipLaunchKernelGGL((conv3x3_wstat_kernel<NRM, 1>), grid, dim3(256), 0, s, a);
  else if (kind == 3) hipLaunchKernelGGL((conv3x3_wstat_kernel<NRM, 3>), grid, dim3(256), 0, s, a);
#ifdef PWR_DEBUG_BUILD      // (the norm-backward-sums form: measured and not shipped -- see the header; the product library does not carry it)
  else if constexpr (!NRM) hipLaunchKernelGGL((conv3x3_wstat_kernel<false, 2>), grid, dim3(256), 0, s, a);
#endif
}

// one job (b == nullptr) or two jobs of one geometry, one norm / statistics form
int launch_conv_wstat(const ConvParams& pa, const ConvParams* pb, hipStream_t s) {
  const int ncu = PWR_DBG_ENV("PWR_WSTAT_WGS", 256);
  WstatArgs a;
  a.job[0] = pa;
  a.job[1] = pb ? *pb : pa;
#ifdef PWR_DEBUG_BUILD
  a.job[0].stamps = a.job[1].stamps = wstat_stamps();
#endif
  a.njobs = pb ? 2 : 1;
  const int tiles = pa.B * (pa.H / 4) * (pa.W / 32);
  int per = ncu / a.njobs;
  if (per > tiles) per = tiles;
  a.wgs_per_job = per;
  a.tiles_q = tiles / per; a.tiles_r = tiles % per;
  a.job_shift = pb ? 1 : 0; a.nx_shift = pb ? 2 : 3;
  a.xq = per / (8 >> a.job_shift); a.xr = per % (8 >> a.job_shift);
  const unsigned tiles_x = (unsigned)(pa.W / 32), tiles_img = tiles_x * (unsigned)(pa.H / 4);
  a.magic_img = tiles_img > 1 ? (unsigned)(((1ull << 32) + tiles_img - 1) / tiles_img) : 0xFFFFFFFFu;
  a.magic_x = tiles_x > 1 ? (unsigned)(((1ull << 32) + tiles_x - 1) / tiles_x) : 0xFFFFFFFFu;
  const int kind = pa.y_nchw ? 3 : (pa.st_partial ? 1 : (pa.nb_partial ? 2 : 0));
  dim3 grid(per * a.njobs);
  if (pa.in_norm) launch_kind<true>(a, kind, grid, s);
  else launch_kind<false>(a, kind, grid, s);
  return (int)hipGetLastError();
}

bool conv_wstat_narrow_pair_applicable(const ConvParams& a, const ConvParams& b, int dtype) {
  return conv_wstat_narrow_applicable(a, dtype) && conv_wstat_narrow_applicable(b, dtype) && a.B == b.B && a.H == b.H && a.W == b.W &&
         (a.in_norm != nullptr) == (b.in_norm != nullptr) && a.relu_in == b.relu_in && a.B * (a.H / 4) * (a.W / 32) >= 32;
}

bool conv_wstat_pair_applicable(const ConvParams& a, const ConvParams& b, int dtype) {
  return conv_wstat_applicable(a, dtype) && conv_wstat_applicable(b, dtype) && a.B == b.B && a.H == b.H && a.W == b.W && a.Cin == b.Cin &&
         (a.in_norm != nullptr) == (b.in_norm != nullptr) && (a.st_partial != nullptr) == (b.st_partial != nullptr) &&
         (a.nb_partial != nullptr) == (b.nb_partial != nullptr) && a.relu_in == b.relu_in;
}

}  // namespace pwr
