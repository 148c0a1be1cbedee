// 3x3 stride-1 128 -> 128 bf16 convolution (the heads' layers, /root/reference/model.py:54-65 / :103-114, and their data
// gradients), "ping-pong" form of conv_patch.hip's kernel.
//
// Why: in conv3x3_patch_kernel every workgroup does [stage the input patch] -> [36 MFMA steps] -> [epilogue], and all
// workgroups of a launch run these phases in lockstep (per-workgroup s_memtime stamps, tools/stamp_patch.py: 24 % staging,
// 49 % K loop at ~90 % MFMA issue, 27 % epilogue): the chip alternates between a load burst with idle matrix cores, a compute
// phase with idle memory, and a store burst.  Here ONE 8-wave workgroup per CU owns two patches in LDS (135 KB) and works on
// two tiles half a period apart: while the four waves of one group run the K loop of their tile, the other four store the
// previous tile (accumulators -> LDS -> coalesced 16-byte stores, norm statistics) and stage their next one (global -> registers
// -> norm + ReLU -> LDS).  Both groups pass the same workgroup barriers (36 weight-ring steps + 1 hand-over per half period), the
// memory group's work is cut into the intervals between them.
//
// The weight ring (three LDS stages of [128][64 B], filled by LDS-DMA, fragments one step ahead in registers) runs on without a
// break across tiles: the computing group issues the DMA for step it+3 -- the last three of a tile are the first three of the
// next (same weights for every tile) -- and, once it has become the memory group, retires them (counted vmcnt) before the first
// two barriers of the other group's K loop.
#include <cstdlib>

#include "conv_common.h"
#include "pwr.h"

namespace pwr {
namespace {

typedef bf16_t T;
typedef bf16x8 V;
constexpr int CIN = 128, BN = 128, EP = 8, KCH = 4, ITERS = 9 * KCH;
constexpr int TH = 4, TW = 32, PH = TH + 2, PW = TW + 2, PP = PH * PW, NSLOT = CIN / EP, PITCH = NSLOT * 16 + 16;
constexpr int PATCH_BYTES = PP * PITCH;      // 55488
constexpr int WBUF_BYTES = BN * 64;          // 8192
constexpr int EROWS = 64, EPITCH = BN + 4;
constexpr int NIT = (PP + 15) / 16;          // staged pixels per thread (256 threads = 16 pixel lanes x 16 channel slots)
constexpr int RING = 6;                      // weight stages in LDS: a DMA has RING - 1 K steps (~300 clocks each) to land
constexpr int LDS_BYTES = 2 * PATCH_BYTES + RING * WBUF_BYTES;   // 160128 of 163840
static_assert(EROWS * EPITCH * 4 <= PATCH_BYTES, "the epilogue buffer lives in the group's own patch");
static_assert(ITERS % RING == 0, "the ring continues across tiles: stage ITERS must land in buffer 0");

// s_waitcnt immediates (gfx9: vmcnt[3:0] + [15:14], expcnt[6:4], lgkmcnt[11:8]); expcnt = 7 (no wait)
constexpr int WAIT_LGKM0 = 0xC07F;           // lgkmcnt(0) only
constexpr int WAIT_VM0_LGKM0 = 0x0070;       // vmcnt(N) lgkmcnt(0), N < 16: 0x0070 + N
constexpr int WAIT_RING = WAIT_VM0_LGKM0 + 2 * (RING - 2);   // all but the newest RING - 2 stages (2 instructions each) have landed

template <int IMM>
__device__ __forceinline__ void wait_barrier() {
  __atomic_signal_fence(__ATOMIC_SEQ_CST);
  __builtin_amdgcn_s_waitcnt(IMM);
  __builtin_amdgcn_s_barrier();
  __atomic_signal_fence(__ATOMIC_SEQ_CST);
}

__global__ __launch_bounds__(512) void conv3x3_pingpong_kernel(ConvParams p, int tiles_total, int tiles_per_wg) {
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
  const int tid = threadIdx.x, gt = tid & 255, lane = tid & 63;
  // wave-uniform ids as SCALARS: the role branch below must be a scalar branch (as a divergent one both roles' registers
  // would be live at once)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave >> 2, w4 = wave & 3;
  const int wm = w4 >> 1, wn = w4 & 1, r = lane & 31, hh = lane >> 5;
  char* patch = smem + g * PATCH_BYTES;
  char* wbuf = smem + 2 * PATCH_BYTES;
  float* E = reinterpret_cast<float*>(patch);
  const int HW = p.H * p.W, tiles_x = p.W / TW, tiles_img = tiles_x * (p.H / TH);
  const int first = xcd_remap(blockIdx.x, gridDim.x) * tiles_per_wg;
  const int n = min(tiles_per_wg, tiles_total - first);
  if (n <= 0) return;   // (whole workgroup)
  const T* __restrict__ x = reinterpret_cast<const T*>(p.x);
  const char* __restrict__ w = reinterpret_cast<const char*>(p.w);

  // weight stage `stage` (= tap * KCH + kch) -> ring buffer `buf`: two 1-KiB LDS-DMA instructions per wave of the issuing group
  // (row = 16 ch + lane / 4 and the source-side XOR swizzle (lane & 3) ^ ((row >> 2) & 3) = (lane & 3) ^ ((lane >> 4) & 3) do not
  // depend on the stage or on i: one 32-bit lane offset, everything else is wave-uniform -> scalar base + VGPR offset addressing)
  const unsigned dma_lane_off = (unsigned)((16 * w4 + (lane >> 2)) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) << 4));
  auto dma_w = [&](int stage, int buf, unsigned lane_off) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ch = i * 4 + w4;
      const char* sbase = w + ((size_t)stage * p.CoutPad * 64 + i * 4096);      // wave-uniform
      const char* src = sbase + (size_t)lane_off;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(wbuf + buf * WBUF_BYTES + ch * 1024), 16, 0, 0);
    }
  };
  struct TileXY { int b, tr, ty0, tx0; };
  auto tile_xy = [&](int t) {
    TileXY q;
    q.b = t / tiles_img; q.tr = t - q.b * tiles_img;
    q.ty0 = (q.tr / tiles_x) * TH; q.tx0 = (q.tr % tiles_x) * TW;
    return q;
  };
  // ---- staging: thread -> pixel lane pl (16) x channel slot (16); pixels pl + 16 k of the (TH+2) x (TW+2) patch
  auto stage_load = [&](const TileXY& q, V (&v)[NIT], unsigned& okm, int pl, int slot) {
    const T* __restrict__ xs = x + (size_t)q.b * HW * CIN;
    okm = 0;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int pix = pl + k * 16;
      const int py = pix / PW, px = pix - py * PW;
      const int iy = q.ty0 + py - 1, ix = q.tx0 + px - 1;
      const bool ok = pix < PP && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      v[k] = V{};
      if (ok) { v[k] = *reinterpret_cast<const V*>(xs + ((size_t)iy * p.W + ix) * CIN + slot * EP); okm |= 1u << k; }
    }
  };
  struct NormRegs { float mu[EP], sc[EP], be[EP]; };
  auto stage_norm_load = [&](const TileXY& q, NormRegs& nrm, int slot) {
    if (p.in_norm) {
      const size_t plane = (size_t)p.B * CIN;
      const float* st = p.in_norm + (size_t)q.b * CIN + slot * EP;
#pragma unroll
      for (int e = 0; e < EP; ++e) { nrm.mu[e] = st[e]; nrm.sc[e] = st[2 * plane + e]; nrm.be[e] = st[3 * plane + e]; }
    }
  };
  auto stage_write = [&](const V (&v)[NIT], unsigned okm, const NormRegs& nrm, int k0, int k1, int pl, int slot) {
    const bool nr = p.in_norm != nullptr;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      if (k < k0 || k >= k1) continue;
      const int pix = pl + k * 16;
      if (pix < PP) {
        V o = v[k];
        if (nr && ((okm >> k) & 1)) {
#pragma unroll
          for (int e = 0; e < EP; ++e) {
            float f = fmaf(Elem<T>::to_f(v[k][e]) - nrm.mu[e], nrm.sc[e], nrm.be[e]);
            if (p.relu_in) f = fmaxf(f, 0.f);
            o[e] = Elem<T>::from_f(f);
          }
        }
        *reinterpret_cast<V*>(patch + pix * PITCH + slot * 16) = o;
      }
    }
  };
  // ---- fragment addresses (as in conv_patch.hip)
  const char* aBase[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) aBase[i] = patch + ((wm * 2 + i) * PW + r) * PITCH + hh * 16;
  int bOff[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    bOff[j][0] = lds_off(wn * 64 + j * 32 + r, hh);
    bOff[j][1] = lds_off(wn * 64 + j * 32 + r, 2 + hh);
  }
  auto frag_load = [&](int it, V (&a)[2][2], V (&bq)[2][2]) {
    const int tap = it / KCH, kch = it - tap * KCH;
    const int ky = tap / 3, kx = tap - ky * 3;
    const char* lB = wbuf + (it % RING) * WBUF_BYTES;
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
#pragma unroll
      for (int i = 0; i < 2; ++i) a[ss][i] = *reinterpret_cast<const V*>(aBase[i] + (ky * PW + kx) * PITCH + kch * 64 + ss * 32);
#pragma unroll
      for (int j = 0; j < 2; ++j) bq[ss][j] = *reinterpret_cast<const V*>(lB + bOff[j][ss]);
    }
  };

  // ---- prologue: group 0 primes the ring and stages the first tile
  if (g == 0) {
#pragma unroll
    for (int k = 0; k < RING; ++k) dma_w(k, k, dma_lane_off);
    V sv[NIT]; unsigned okm;
    const TileXY q = tile_xy(first);
    NormRegs nrm;
    stage_load(q, sv, okm, gt >> 4, gt & 15);
    stage_norm_load(q, nrm, gt & 15);
    stage_write(sv, okm, nrm, 0, NIT, gt >> 4, gt & 15);
  }
  __syncthreads();   // (also drains the DMA stages)

  float bias_r[EP];
#pragma unroll
  for (int e = 0; e < EP; ++e) bias_r[e] = p.bias ? p.bias[(gt & 15) * EP + e] : 0.f;
  f32x16 acc[2][2];
  for (int h = 0; h <= n; ++h) {
    if ((h & 1) == g) {
      // =============================================================== compute role: tile first + h
      if (h < n) {
        const bool has_next = h + 1 < n;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        // (opaque per phase: the 72 source addresses are then formed on the fly as scalar base + this offset instead of being
        // precomputed outside the tile loop and spilled)
        unsigned doff = dma_lane_off;
        asm volatile("" : "+v"(doff));
        V fa[2][2][2], fb[2][2][2];
        frag_load(0, fa[0], fb[0]);
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
          // stage it+1 landed (own share), own LDS reads retired.  In flight: the stages it+2 .. it+RING-1 -- of this tile, and of
          // the next one if there is one; at the tail of the last tile only what is left of it
          if (it + RING - 1 < ITERS) wait_barrier<WAIT_RING>();
          else if (has_next) wait_barrier<WAIT_RING>();
          else {
            const int left = ITERS - 2 - it;       // stages that may still be in flight (constant after unrolling)
            static_assert(RING == 6, "tail waits written out for RING - 2 = 4 stages");
            if (left >= 3) wait_barrier<WAIT_VM0_LGKM0 + 6>();
            else if (left == 2) wait_barrier<WAIT_VM0_LGKM0 + 4>();
            else if (left == 1) wait_barrier<WAIT_VM0_LGKM0 + 2>();
            else wait_barrier<WAIT_VM0_LGKM0>();
          }
          if (p.stamps && (it % 6 == 0 || it == ITERS - 1) && w4 == 0 && lane == 0 && h < 16)
            p.stamps[((size_t)blockIdx.x * 16 + h) * 8 + (it == ITERS - 1 ? 6 : it / 6)] = (long long)__builtin_amdgcn_s_memtime();
          if (it + RING < ITERS) dma_w(it + RING, it % RING, doff);
          else if (has_next) dma_w(it + RING - ITERS, it % RING, doff);
          if (it + 1 < ITERS) frag_load(it + 1, fa[(it + 1) & 1], fb[(it + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int ss = 0; ss < 2; ++ss)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[it & 1][ss][i], fb[it & 1][ss][j], acc[i][j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        wait_barrier<WAIT_LGKM0>();           // hand-over
      } else {
#pragma unroll 1
        for (int k = 0; k <= ITERS; ++k) wait_barrier<WAIT_LGKM0>();
      }
    } else {
      // =============================================================== memory role: store tile first + h - 1, stage tile first + h + 1
      const bool do_epi = h >= 1, do_stage = h + 1 < n;
      // opaque copy of the thread index: keeps the address arithmetic of this role INSIDE the role (hoisted out of the tile
      // loop it would stay live across the other role's K loop, whose registers are all spoken for)
      int gtv = gt;
      asm volatile("" : "+v"(gtv));
      const int pl = gtv >> 4, slot = gtv & 15;
      TileXY qe = {0, 0, 0, 0}, qs = {0, 0, 0, 0};
      if (do_epi) qe = tile_xy(first + h - 1);
      if (do_stage) qs = tile_xy(first + h + 1);
      // The memory role never waits for memory inside an interval (the other group's K loop would wait at the barrier with it):
      // every load is issued at least two intervals before its first use.
      EpiStats<T> est;
      T* __restrict__ y = reinterpret_cast<T*>(p.y);
      const int kind = p.st_partial ? 1 : (p.nb_partial ? 2 : 0);
      V yv[8];                                      // kind 2: the forward activations of this thread's 8 output vectors
      auto out_row = [&](int ps, int qn) {          // NHWC pixel index of this thread's vector in pass ps, quarter qn
        const int ml = ps * EROWS + ((gtv + 256 * qn) >> 4);
        return (size_t)qe.b * HW + (size_t)(qe.ty0 + ml / TW) * p.W + qe.tx0 + ml % TW;
      };
      auto write_E = [&](int ps) {
        if (wm == ps) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int e = 0; e < 16; ++e) {
                const int row = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                E[row * EPITCH + wn * 64 + j * 32 + r] = acc[i][j][e];
              }
        }
      };
      auto copy_chunk = [&](int ps, int qn) {       // rows of pass ps, quarter qn: one 16-byte vector per thread
        const int row = (gtv + 256 * qn) >> 4, cc = slot * EP;
        const size_t m = out_row(ps, qn);
        V o;
#pragma unroll
        for (int e = 0; e < EP; ++e) o[e] = Elem<T>::from_f(E[row * EPITCH + cc + e] + bias_r[e]);
        *reinterpret_cast<V*>(y + m * p.Cout + cc) = o;
        if (kind == 1) {
#pragma unroll
          for (int e = 0; e < EP; ++e) { const float d = Elem<T>::to_f(o[e]) - est.a0[e]; est.s1[e] += d; est.s2[e] = fmaf(d, d, est.s2[e]); }
        } else if (kind == 2) {
          const V yq = yv[ps * 4 + qn];
#pragma unroll
          for (int e = 0; e < EP; ++e) {
            const float yy = Elem<T>::to_f(yq[e]);
            float gg = Elem<T>::to_f(o[e]);
            if (p.nb_relu && !(fmaf(yy - est.a0[e], est.a2[e], est.a3[e]) > 0.f)) gg = 0.f;
            est.s1[e] += gg;
            est.s2[e] = fmaf(gg, (yy - est.a0[e]) * est.a1[e], est.s2[e]);
          }
        }
      };
      // I_0 .. : (this group's DMA stages for the other group's first RING - 1 steps retire one per barrier; no other
      // vector-memory operation of these waves before the last of them)
      if (do_epi) write_E(0);
      wait_barrier<WAIT_VM0_LGKM0 + 8>();                                   // B_0
      wait_barrier<WAIT_VM0_LGKM0 + 6>();                                   // B_1
      wait_barrier<WAIT_VM0_LGKM0 + 4>();                                   // B_2
      wait_barrier<WAIT_VM0_LGKM0 + 2>();                                   // B_3
      wait_barrier<WAIT_VM0_LGKM0>();                                       // B_4
      if (do_epi) {
        est.init(p, qe.b, slot * EP);                                       // (kind 2: loads the norm state of the sample)
        if (kind == 2) {
#pragma unroll
          for (int u = 0; u < 8; ++u) yv[u] = *reinterpret_cast<const V*>(reinterpret_cast<const T*>(p.nb_y) + out_row(u >> 2, u & 3) * p.Cout + slot * EP);
        }
      }
      wait_barrier<WAIT_LGKM0>();                                           // B_5
      wait_barrier<WAIT_LGKM0>();                                           // B_6
      wait_barrier<WAIT_LGKM0>();                                           // B_7
      if (do_epi) {
        if (kind == 1) {
#pragma unroll
          for (int e = 0; e < EP; ++e) est.a0[e] = E[slot * EP + e] + bias_r[e];     // the shift: the tile's first output row
        }
        copy_chunk(0, 0);
      }
      wait_barrier<WAIT_LGKM0>();                                           // B_8
      if (do_epi) copy_chunk(0, 1);
      wait_barrier<WAIT_LGKM0>();                                           // B_9
      if (do_epi) copy_chunk(0, 2);
      wait_barrier<WAIT_LGKM0>();                                           // B_10
      if (do_epi) copy_chunk(0, 3);
      wait_barrier<WAIT_LGKM0>();                                           // B_11
      if (do_epi) write_E(1);
      wait_barrier<WAIT_LGKM0>();                                           // B_12
      if (do_epi) copy_chunk(1, 0);
      wait_barrier<WAIT_LGKM0>();                                           // B_13
      if (do_epi) copy_chunk(1, 1);
      wait_barrier<WAIT_LGKM0>();                                           // B_14
      if (do_epi) copy_chunk(1, 2);
      wait_barrier<WAIT_LGKM0>();                                           // B_15
      if (do_epi) copy_chunk(1, 3);
      wait_barrier<WAIT_LGKM0>();                                           // B_16
      // column statistics of the tile (EpiStats::finish, split at its barrier); 16 slots x EP channels, 4 waves
      if (do_epi && kind != 0) {
#pragma unroll
        for (int o = 16; o < 64; o <<= 1)
#pragma unroll
          for (int e = 0; e < EP; ++e) { est.s1[e] += __shfl_xor(est.s1[e], o, 64); est.s2[e] += __shfl_xor(est.s2[e], o, 64); }
        if (lane < 16) {
#pragma unroll
          for (int e = 0; e < EP; ++e) { E[((w4 * 2 + 0) * 16 + slot) * EP + e] = est.s1[e]; E[((w4 * 2 + 1) * 16 + slot) * EP + e] = est.s2[e]; }
        }
      }
      wait_barrier<WAIT_LGKM0>();                                           // B_17
      if (do_epi && kind != 0) {
        float* out = kind == 1 ? p.st_partial + ((size_t)(qe.b * tiles_img + qe.tr) * 3) * p.Cout
                               : p.nb_partial + ((size_t)(qe.b * tiles_img + qe.tr) * 2) * p.Cout;
        const int which = gtv >> 7, c = gtv & 127;
        float t = 0.f;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) t += E[((wv * 2 + which) * 16) * EP + c];
        out[(size_t)which * p.Cout + c] = t;
        if (kind == 1 && gtv < 16) {
#pragma unroll
          for (int e = 0; e < EP; ++e) out[(size_t)2 * p.Cout + slot * EP + e] = est.a0[e];
        }
      }
      wait_barrier<WAIT_LGKM0>();                                           // B_18
      V sv[NIT]; unsigned okm = 0;
      NormRegs nrm;
      if (do_stage) { stage_load(qs, sv, okm, pl, slot); stage_norm_load(qs, nrm, slot); }
      wait_barrier<WAIT_LGKM0>();                                           // B_19
#pragma unroll 1
      for (int k = 20; k < 32; ++k) wait_barrier<WAIT_LGKM0>();             // B_20 .. B_31 (the loads are in flight)
      if (do_stage) stage_write(sv, okm, nrm, 0, 4, pl, slot);
      wait_barrier<WAIT_LGKM0>();                                           // B_32
      if (do_stage) stage_write(sv, okm, nrm, 4, 7, pl, slot);
      wait_barrier<WAIT_LGKM0>();                                           // B_33
      if (do_stage) stage_write(sv, okm, nrm, 7, 10, pl, slot);
      wait_barrier<WAIT_LGKM0>();                                           // B_34
      if (do_stage) stage_write(sv, okm, nrm, 10, NIT, pl, slot);
      wait_barrier<WAIT_LGKM0>();                                           // B_35
      wait_barrier<WAIT_LGKM0>();                                           // hand-over
    }
  }
}

}  // namespace

static int g_pingpong_override = -1;
void set_debug_pingpong(int v) { g_pingpong_override = v; }

bool conv_pingpong_applicable(const ConvParams& p, int dtype) {
  // Measured on MI355X (tools/test_pingpong.py, tools/stamp_pp.py; bitwise equal to conv3x3_patch_kernel on every shape tried):
  // 111 us vs 50 us at B=32 (4 tiles per CU), 198 vs 93 us at B=64.  A K step takes ~405 clocks while the memory group has
  // nothing to do, but 700 - 2500 in the intervals where it works: role switching costs ~100 scratch reloads per half period
  // (256 registers are not enough for both roles' invariants), and a scratch reload waits for every older store of the wave.
  // The schedule itself also pays one fill / drain half period per launch (5 for 4 tiles).  Opt-in until the roles are split
  // into dedicated waves (no switching) -- next round.
  static const bool env_on = [] { const char* e = getenv("PWR_PINGPONG"); return e ? atoi(e) != 0 : false; }();
  const bool on = g_pingpong_override < 0 ? env_on : g_pingpong_override != 0;
  if (!on || dtype != PWR_BF16) return false;
  if (!(p.mode == 0 && p.ksize == 3 && p.stride == 1 && p.pad == 1 && p.Cin == CIN && p.Cout == BN && p.CoutPad == BN)) return false;
  if (p.W % TW || p.H % TH || p.residual || p.y_nchw || !p.y) return false;
  const long long tiles = (long long)p.B * (p.H / TH) * (p.W / TW);
  return tiles >= 1024;   // at least four tiles per CU: below that there is nothing to alternate with
}

static long long* g_pp_stamps = nullptr;
void set_debug_stamps_pp(long long* ptr) { g_pp_stamps = ptr; }

int launch_conv_pingpong(const ConvParams& p0, hipStream_t s) {
  ConvParams p = p0;
  p.stamps = g_pp_stamps;
  const int tiles = p.B * (p.H / TH) * (p.W / TW);
  static const int wgs = [] { const char* e = getenv("PWR_PINGPONG_WGS"); return e ? atoi(e) : 256; }();
  const int per = (tiles + wgs - 1) / wgs;
  const int grid = (tiles + per - 1) / per;
  hipLaunchKernelGGL(conv3x3_pingpong_kernel, dim3(grid), dim3(512), 0, s, p, tiles, per);
  return (int)hipGetLastError();
}

}  // namespace pwr
