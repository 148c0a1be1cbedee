// 3x3 stride-1 convolution with the input patch resident in LDS (the hot kernel: the heads' 128->128 convs are 72 % of
// the conv FLOPs of /root/reference/model.py:54-65 / :103-114; also the stem's 32->64 / 64->128 convs :171-179, the
// hourglass 64->64 convs :16 at 64x64 / 32x32, and every data gradient of those, which is the same conv with flipped weights).
//
// Workgroup = 256 threads, output tile = 4 rows x 32 columns (or 512 threads, 8 x 32) x BN channels.  The (TH+2) x 34 pixel input
// patch (tile + halo) is loaded ONCE from HBM/L2 -- with the preceding norm + ReLU applied on the way ("NR prologue") --
// into LDS as [pixel][Cin] (pixel pitch padded by 16 B), and all nine taps read their shifted windows from
// there: 1.6x input traffic instead of the 9x of the universal im2col-on-the-fly kernel, and the NR arithmetic is done
// 1.6x instead of 9x.  Weights stream through a 3-stage LDS-DMA ring of [BN][64 B] tiles per (tap, K chunk); they are shared
// by all workgroups and stay in L2.  MFMA 32x32x16 bf16 (or 32x32x2 fp32 in parity mode), fp32 accumulate.
//
// Small square maps (TW = W = H in {2,4,8,16}: the inner hourglass levels) use the same kernel with the 128-pixel tile made
// of SUB sub-blocks of RH x TW pixels, each with its own halo patch -- whole images (or half of a 16x16 one) of several
// samples.  There the point is latency, not traffic: one round of global loads instead of one per (tap, K chunk).
#include <cstdlib>

#include "conv_common.h"
#include "pwr.h"

namespace pwr {

#ifdef PWR_DEBUG_BUILD
static long long* g_stamps = nullptr;     // (pwr_debug_set_stamps)
static int g_delay = PWR_DBG_ENV("PWR_PATCH_DELAY", 0);      // (pwr_debug_set_delay / the environment)
#else
static constexpr long long* g_stamps = nullptr;     // the shipped library has neither the stamps nor the delay, nor their setters
static constexpr int g_delay = 0;
#endif
__device__ __forceinline__ void stamp(const ConvParams& p, int slot) {
#ifdef PWR_DEBUG_BUILD
  if (p.stamps && threadIdx.x == 0) {
    long long* d = p.stamps + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8;
    d[slot] = (long long)__builtin_amdgcn_s_memtime();
    if (slot == 0) { d[6] = __builtin_amdgcn_s_getreg((31 << 11) | 4); d[7] = __builtin_amdgcn_s_getreg((31 << 11) | 20); }
  }
#endif
}

// Patch pixels are padded by one 16-byte slot: with a pitch of NSLOT*16+16 bytes the 16-lane groups of ds_read_b128
// (32 consecutive pixels, same channel slot) fall on 16 distinct bank slots WITHOUT an XOR swizzle, so every fragment
// address is (per-lane base) + (compile-time constant): no vector ALU work in the K loop for the A operand.
// KS = 1: the same kernel as a 1x1 convolution (no halo, taps = 1): the whole K extent of a tile's input is then resident after
// ONE round of loads (8 per thread, all in flight together) -- the universal kernel exposes a global-load round trip per K step,
// which for these memory-bound layers (the ResBlocks' bottleneck convs, the stage-input conv) meant ~2 TB/s.
// GEO = 2 + 2 py + px: the data gradient of a STRIDE-2 3x3 conv for the output pixels of parity (py, px): dx[2a+py, 2c+px] reads
// dy[a + oy, c + ox] for 1, 2, 2 or 4 taps (py = 0: ky = 1; py = 1: ky = 2 at oy = 0 and ky = 0 at oy = 1; same in x) -- a stride-1
// patch conv over dy with a (1+py) x (1+px) window whose output is scattered to every second pixel.  Four classes (one launch since round 6:
// conv3x3_patch_tr4_kernel; four launches before)
// replace the gather form of the universal kernel (180 us isolated for the stem's 128 <- 128 layer at 128x128).
// GEO = 6: the FORWARD of a stride-2 3x3 conv (the stem's last layer, model.py:182): y[oy, ox] reads x[2 oy + ky - 1, 2 ox + kx - 1], i.e.
// tap (ky, kx) reads the input pixels of parity class (ky != 1, kx != 1), which form a stride-1 image of their own.  The workgroup walks
// the four classes one after the other -- stage the class's (TH+1) x 33 window of the tile (every second pixel of the input, norm +
// ReLU on the way), run its 1, 2, 2 or 4 taps out of LDS, next class -- into ONE set of accumulators; the weight ring runs through.
// 9 taps x KCH K steps like the stride-1 conv, 4 x 165 staged pixels per 128 outputs (the universal implicit-GEMM kernel, which
// gathered 9 x 128 pixels per tile with a global-load round trip per K step, took 138 us for the FLOPs of a 43-us head conv).
// MF = 16: the K loop issues v_mfma_f32_16x16x32_bf16 instead of 32x32x16 (same cycles per FLOP, same LDS reads per FLOP, same 64
// accumulator registers for a 64 x 64 wave tile: 16 tiles of 4).  The chip holds its clock down under this kernel (all-zero operands:
// 38.8 vs 47 - 49 us, profiles/r3_experiments.md section 6), and the clock it holds depends on the MFMA shape (MI355X_MICROARCH.md,
// DVFS give-back item 7: build both at the same tile, keep the faster by wall on random data).
__host__ __device__ constexpr int s2_tap(int q) { return q == 0 ? 4 : q == 1 ? 3 : q == 2 ? 5 : q == 3 ? 1 : q == 4 ? 7 : q == 5 ? 0 : q == 6 ? 2 : q == 7 ? 6 : 8; }
__host__ __device__ constexpr bool s2_first(int q) { return q == 0 || q == 1 || q == 3 || q == 5; }       // first tap position of a class
// Patch + weight-ring (or epilogue image) bytes of one variant of the kernel: conv3x3_patch_body's own LDS array has this size; a kernel that
// holds SEVERAL variants (conv3x3_patch_tr4_kernel) declares one array of the largest and hands it in (EXT)
template <typename T, int CIN, int WM, int WN, int MR, int NR, bool DMA, int TW, int GEO, bool FB = false>
__host__ __device__ constexpr int patch_lds_bytes() {
  constexpr int EP = Mma<T>::EP, KE = Mma<T>::KE, NT = WM * WN * 64, BM = WM * MR * 32, BN = WN * NR * 32, TH = WM * MR;
  constexpr int RH = TW == 32 ? TH : (TW * TW < BM ? TW : BM / TW), SUB = BM / (RH * TW);
  constexpr bool TR = GEO >= 2 && GEO <= 5, S2 = GEO == 6;
  constexpr int CPY = TR ? ((GEO - 2) >> 1) : 0, CPX = TR ? ((GEO - 2) & 1) : 0, HALO = GEO == 0 ? 1 : 0;
  constexpr int PH = RH + (TR ? CPY : (S2 ? 1 : 2 * HALO)), PW = TW + (TR ? CPX : (S2 ? 1 : 2 * HALO));
  constexpr int PATCH = SUB * PH * PW * ((CIN / EP) * 16 + 16), WB = BN * 64, EPI = 64 * (BN + 4) * 4;
  constexpr int NSTAGE = DMA ? (GEO == 1 ? CIN / KE : 3) : 2;
  (void)NT;
  return ((PATCH + NSTAGE * WB) > EPI ? (PATCH + NSTAGE * WB) : EPI) + (FB ? 2 * CIN * 4 : 0);     // (FB: the 2 Cin sums of the fold behind everything)
}
template <typename T, int CIN, int WM, int WN, int MR, int NR, bool DMA, int TW = 32, int GEO = 0, int MF = 32, bool EXT = false, bool FB = false>
#ifndef PWR_OCC_HINT
#define PWR_OCC_HINT 1
#endif
__device__ __forceinline__ void conv3x3_patch_body(const ConvParams& p, char* ext_smem = nullptr) {
  typedef typename Vec16<T>::type V;
  static_assert(MF == 32 || (MF == 16 && sizeof(T) == 2 && DMA), "the 16x16x32 form exists for the bf16 LDS-DMA kernel");
  constexpr int MR4 = MR * 2, NR4 = NR * 2;     // 16-row / 16-column blocks of the wave tile (MF == 16)
  constexpr int KE = Mma<T>::KE, EP = Mma<T>::EP;
  constexpr int NT = WM * WN * 64;                // threads per workgroup (256 or 512)
  constexpr int BM = WM * MR * 32, BN = WN * NR * 32;
  constexpr int TH = WM * MR;                     // tile rows of the TW == 32 form
  constexpr int RH = TW == 32 ? TH : (TW * TW < BM ? TW : BM / TW);   // rows of one sub-block
  constexpr int SUBPIX = RH * TW, SUB = BM / SUBPIX;                  // 128-pixel tile = SUB sub-blocks of RH x TW pixels
  constexpr bool TR = GEO >= 2 && GEO <= 5;
  constexpr bool S2 = GEO == 6;
  static_assert(!S2 || (DMA && TW == 32), "the stride-2 forward form: bf16 LDS-DMA kernel, 4 x 32 tile");
  constexpr int CPY = TR ? ((GEO - 2) >> 1) : 0, CPX = TR ? ((GEO - 2) & 1) : 0;
  constexpr int HALO = GEO == 0 ? 1 : 0;
  constexpr int NTY = TR ? CPY + 1 : ((GEO == 0 || S2) ? 3 : 1), NTX = TR ? CPX + 1 : ((GEO == 0 || S2) ? 3 : 1);     // taps per axis
  constexpr int PH = RH + (TR ? CPY : (S2 ? 1 : 2 * HALO)), PW = TW + (TR ? CPX : (S2 ? 1 : 2 * HALO)), PP = PH * PW, NPIX = SUB * PP;
  static_assert(!TR || TW == 32, "the transposed classes use the 4 x 32 tile");
  // S2: K steps in class order -- tap position q -> kernel tap ky * 3 + kx; class 0: (1,1); 1: (1,0) (1,2); 2: (0,1) (2,1); 3: the corners
  // (s2_tap / s2_first above the kernel)
  constexpr int NSLOT = CIN / EP;                 // 16-byte slots per pixel
  constexpr int KCH = CIN / KE;                   // 64-byte K chunks per tap
  // weight stage (tap-major index into the pack) of K step `it`; the window offset of its tap inside the patch is (it's ty, tx)
  auto wstage = [](int it) {
    const int tap = it / KCH, kch = it - tap * KCH;
    const int ty = tap / NTX, tx = tap - ty * NTX;
    int wt;
    if (GEO == 0) wt = ty * 3 + tx;
    else if (GEO == 1) wt = 0;
    else if (S2) wt = s2_tap(tap);
    else wt = (CPY ? (ty == 0 ? 2 : 0) : 1) * 3 + (CPX ? (tx == 0 ? 2 : 0) : 1);
    return wt * KCH + kch;
  };
  constexpr int PITCH = NSLOT * 16 + 16;         // bytes per patch pixel (padded, see above)
  constexpr int PATCH_BYTES = NPIX * PITCH;
  constexpr int WBUF_BYTES = BN * 64;
  constexpr int NB = (BN * 4 + NT - 1) / NT;
  constexpr int EROWS = 64, EPITCH = BN + 4;
  constexpr int EPI_BYTES = EROWS * EPITCH * 4;
  constexpr int NSTAGE = DMA ? (GEO == 1 ? KCH : 3) : 2;   // weight ring: LDS-DMA runs two K steps ahead (1x1: ALL <= 4 stages resident)
  constexpr int LDS_MAIN = (PATCH_BYTES + NSTAGE * WBUF_BYTES) > EPI_BYTES ? (PATCH_BYTES + NSTAGE * WBUF_BYTES) : EPI_BYTES;
  constexpr int LDS_BYTES = LDS_MAIN + (FB ? 2 * CIN * 4 : 0);
  static_assert(LDS_BYTES == patch_lds_bytes<T, CIN, WM, WN, MR, NR, DMA, TW, GEO, FB>(), "patch_lds_bytes() out of step with the kernel body");
  static_assert(!FB || (GEO == 0 && TW == 32 && sizeof(T) == 2 && NT == 2 * CIN), "the fold form: bf16 3x3 stride-1 tiles, one thread per sum");
  __shared__ __attribute__((aligned(16))) char smem_own[EXT ? 16 : LDS_BYTES];
  char* smem = EXT ? ext_smem : smem_own;
  char* patch = smem;
  char* wbuf = smem + PATCH_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int t = xcd_remap(blockIdx.x, gridDim.x);
  const int HW = p.H * p.W;
  const int OW = S2 ? p.Wo : p.W, OH = S2 ? p.Ho : p.H, OHW = OW * OH;      // the image the tiles cover (S2: the output, half the input)
  int b = 0, ty0 = 0, tx0 = 0, tr = 0, tiles_img = 1;
  if constexpr (TW == 32) {
    const int tiles_x = OW / TW, tiles_y = OH / TH;   // requires H % TH == 0
    tiles_img = tiles_x * tiles_y;
    b = t / tiles_img;
    tr = t - b * tiles_img;
    ty0 = (tr / tiles_x) * TH; tx0 = (tr % tiles_x) * TW;
  }
  const long long L0 = (long long)t * BM, Mtot = (long long)p.B * HW;   // small maps: tile = BM consecutive NHWC pixels
  const int n0 = blockIdx.y * BN;
  const T* __restrict__ x = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ w = reinterpret_cast<const T*>(p.w);
  stamp(p, 0);
#ifdef PWR_DEBUG_BUILD
  // phase experiment: the two workgroups of a CU start together and stay in phase (staging beside staging, K loop beside K loop);
  // hold the one in the odd wave slots back by ~dbg_delay cycles so that its memory phases fall beside the other's K loop
  if (p.dbg_delay > 0 && (__builtin_amdgcn_s_getreg((4 << 11) | 4) & 1)) {      // HW_ID[3:0] = wave slot
    for (int i = 0; i < p.dbg_delay; i += 1024) __builtin_amdgcn_s_sleep(16);      // (s_sleep n = 64 n cycles)
  }
#endif

  // ---- weights of iteration 0 in flight while the patch is staged
  V rb[NB];
  auto load_w = [&](int it) {
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int s = tid + NT * i;
      if (BN * 4 >= NT * (i + 1) || s < BN * 4)
        rb[i] = *reinterpret_cast<const V*>(w + ((size_t)it * p.CoutPad + n0 + (s >> 2)) * KE + (s & 3) * EP);
    }
  };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int s = tid + NT * i;
      if (BN * 4 >= NT * (i + 1) || s < BN * 4) *reinterpret_cast<V*>(wbuf + buf * WBUF_BYTES + lds_off(s >> 2, s & 3)) = rb[i];
    }
  };
  // LDS-DMA form: global_load_lds writes 64 lanes x 16 B = 1 KiB (16 tile rows) per wave instruction, linearly; the
  // XOR swizzle of the tile is applied to the per-lane SOURCE address instead (cdna_hip_programming.md rule 21).
  constexpr int NCHUNK = WBUF_BYTES / 1024;       // wave instructions per stage
  constexpr int NWAVE = NT / 64;
  constexpr int NBW = (NCHUNK + NWAVE - 1) / NWAVE;   // per wave
  auto dma_w = [&](int it, int stage) {
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
      const int ch = i * NWAVE + wid;
      if (NCHUNK >= NWAVE * (i + 1) || ch < NCHUNK) {
        const int row = 16 * ch + (lane >> 2);
        const int slot = (lane & 3) ^ ((row >> 2) & 3);
        const char* src = reinterpret_cast<const char*>(w) + ((size_t)it * p.CoutPad + n0 + row) * 64 + slot * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(wbuf + stage * WBUF_BYTES + ch * 1024), 16, 0, 0);
      }
    }
  };
  if constexpr (DMA) {
    if constexpr (GEO == 1) {
      // 1x1: the whole K extent of the weights (<= 4 stages) goes to LDS now; the K loop then has no DMA, no waits, no barriers
#pragma unroll
      for (int k = 0; k < KCH; ++k) dma_w(wstage(k), k);
    } else {
      dma_w(wstage(0), 0);
      if (1 < NTY * NTX * KCH) dma_w(wstage(1), 1);
    }
  } else {
    load_w(wstage(0));
  }

  // ---- stage the patch: thread -> fixed sub-block and 16-byte channel slot, pixels pl + k*PL of that sub-block
  constexpr int TPS = NT / SUB, PL = TPS / NSLOT;
  static_assert(TPS % NSLOT == 0 && PL >= 1, "sub-block needs at least one thread per channel slot");
  constexpr int NITP = (PP + PL - 1) / PL;
  const int st_sub = tid / TPS, st_pl = (tid % TPS) / NSLOT, st_slot = tid % NSLOT;
  int sb = b, sy0 = ty0, sx0 = tx0;
  bool sv = true;
  if constexpr (TW != 32) {
    const long long Ls = L0 + st_sub * SUBPIX;
    sv = Ls < Mtot;
    sb = sv ? (int)(Ls / HW) : 0;
    sy0 = ((int)(Ls - (long long)sb * HW)) / TW; sx0 = 0;
  }
  const T* __restrict__ xs = x + (size_t)sb * HW * CIN;
  const bool nr = p.in_norm != nullptr;
  // cls (S2 only): parity class 2 * py + px of the input pixels to stage -- patch pixel (ry, rx) = input (2 (ty0 - 1 + ry) + py,
  // 2 (tx0 - 1 + rx) + px); row / column -1 of the class image is the conv's zero padding
  auto stage_patch = [&](int cls) {
    float mu[EP], sc[EP], be[EP];          // (S2: re-read per class rather than kept alive through the K loop)
    {
      // (branch-free as well: without a norm the three vectors come from the weight pack and are not used)
      const size_t plane = nr ? (size_t)p.B * CIN : 0;
      const float* st = nr ? p.in_norm + (size_t)sb * CIN + st_slot * EP : reinterpret_cast<const float*>(p.w);
#pragma unroll
      for (int e = 0; e < EP; ++e) { mu[e] = st[e]; sc[e] = st[2 * plane + e]; be[e] = st[3 * plane + e]; }
    }
    // (S2 restages inside the K loop with the accumulators live: two rounds of loads instead of one, half the staging registers)
    constexpr int CH = S2 ? (NITP + 1) / 2 : NITP;
#pragma unroll
    for (int k0 = 0; k0 < NITP; k0 += CH) {
      V v[CH];
      bool ok[CH];
#pragma unroll
      for (int kk = 0; kk < CH; ++kk) {
        const int k = k0 + kk;
        const int pix = st_pl + k * PL;
        const int py = pix / PW, px = pix - py * PW;
        int iy, ix;
        if constexpr (S2) {
          const int a = sy0 - 1 + py, c = sx0 - 1 + px;
          iy = 2 * a + (cls >> 1); ix = 2 * c + (cls & 1);
          ok[kk] = k < NITP && pix < PP && a >= 0 && c >= 0;
        } else {
          iy = sy0 + py - HALO; ix = sx0 + px - HALO;
          ok[kk] = sv && pix < PP && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        }
        // branch-free: an out-of-image pixel reads a clamped (valid) address and is zeroed below -- all loads issue back to back
        const int cy = min(max(iy, 0), p.H - 1), cx = min(max(ix, 0), p.W - 1);
        v[kk] = *reinterpret_cast<const V*>(xs + ((size_t)cy * p.W + cx) * CIN + st_slot * EP);
      }
#pragma unroll
      for (int kk = 0; kk < CH; ++kk) {
        const int k = k0 + kk;
        const int pix = st_pl + k * PL;
        if (k < NITP && pix < PP) {
          V o = v[kk];
          if (nr) {
#pragma unroll
            for (int e = 0; e < EP; ++e) {
              float f = fmaf(Elem<T>::to_f(v[kk][e]) - mu[e], sc[e], be[e]);
              if (p.relu_in) f = fmaxf(f, 0.f);
              o[e] = Elem<T>::from_f(f);
            }
          }
          if (!ok[kk]) o = V{};
          *reinterpret_cast<V*>(patch + (st_sub * PP + pix) * PITCH + st_slot * 16) = o;
        }
      }
    }
  };
  // FB: the norm backward of the tensor this conv's input gradient belongs to, applied while the patch is staged (ConvParams::fb_*).
  // norm_bwd_apply_fold_body's arithmetic in its order: every thread sums one of the 2 Cin slab columns of the sample (all loads in flight,
  // k ascending, / HW), the sums cross LDS, then dy = scale * (g m - S1 - xhat S2) per element.  The tile's own 4 x 32 pixels of dy go out
  // to fb_dy (each pixel is the interior of exactly one tile); halo pixels are computed again by the neighbour, out-of-image ones are zero.
  auto stage_patch_fold = [&]() {
    float* ssum = reinterpret_cast<float*>(smem + LDS_MAIN);
    const T* __restrict__ fy = reinterpret_cast<const T*>(p.fb_y) + (size_t)sb * HW * CIN;
    T* __restrict__ fdy = reinterpret_cast<T*>(p.fb_dy) + (size_t)sb * HW * CIN;
    float sacc = 0.f;
    {
      const int qq = tid >= CIN ? 1 : 0, c = tid - qq * CIN;
      const float* pp = p.fb_partial + ((size_t)sb * p.fb_pchunks * 2 + qq) * CIN + c;
      const size_t st = (size_t)2 * CIN;
      int k = 0;
      for (; k + 32 <= p.fb_pchunks; k += 32) {
        float a[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) a[u] = pp[(size_t)(k + u) * st];
#pragma unroll
        for (int u = 0; u < 32; ++u) sacc += a[u];
      }
      for (; k + 8 <= p.fb_pchunks; k += 8) {
        float a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = pp[(size_t)(k + u) * st];
#pragma unroll
        for (int u = 0; u < 8; ++u) sacc += a[u];
      }
      for (; k < p.fb_pchunks; ++k) sacc += pp[(size_t)k * st];
    }
    float mu[EP], rs[EP], sc[EP], sh[EP];
    {
      const size_t plane = (size_t)p.B * CIN;
      const float* st = p.fb_state + (size_t)sb * CIN + st_slot * EP;
#pragma unroll
      for (int e = 0; e < EP; ++e) { mu[e] = st[e]; rs[e] = st[plane + e]; sc[e] = st[2 * plane + e]; sh[e] = st[3 * plane + e]; }
    }
    V gv[NITP], yv[NITP];
    bool ok[NITP];
#pragma unroll
    for (int k = 0; k < NITP; ++k) {
      const int pix = st_pl + k * PL;
      const int py = pix / PW, px = pix - py * PW;
      const int iy = sy0 + py - HALO, ix = sx0 + px - HALO;
      ok[k] = pix < PP && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      const int cy = min(max(iy, 0), p.H - 1), cx = min(max(ix, 0), p.W - 1);
      const size_t o = ((size_t)cy * p.W + cx) * CIN + st_slot * EP;
      gv[k] = *reinterpret_cast<const V*>(xs + o);
      yv[k] = *reinterpret_cast<const V*>(fy + o);
    }
    ssum[tid] = sacc / (float)HW;
    __syncthreads();
    float s1[EP], s2[EP];
#pragma unroll
    for (int e = 0; e < EP; ++e) { s1[e] = ssum[st_slot * EP + e]; s2[e] = ssum[CIN + st_slot * EP + e]; }
    const bool writer = blockIdx.y == 0;
#pragma unroll
    for (int k = 0; k < NITP; ++k) {
      const int pix = st_pl + k * PL;
      if (pix < PP) {
        const int py = pix / PW, px = pix - py * PW;
        V o;
#pragma unroll
        for (int e = 0; e < EP; ++e) {
          const float yy = Elem<T>::to_f(yv[k][e]);
          float gg = Elem<T>::to_f(gv[k][e]);
          if (p.fb_relu && !(fmaf(yy - mu[e], sc[e], sh[e]) > 0.f)) gg = 0.f;
          const float xn = (yy - mu[e]) * rs[e];
          const float r = sc[e] * (gg - s1[e] - xn * s2[e]);     // (norm_bwd_apply_body's expression)
          o[e] = Elem<T>::from_f(r);
        }
        if (!ok[k]) o = V{};
        *reinterpret_cast<V*>(patch + pix * PITCH + st_slot * 16) = o;
        if (writer && py >= HALO && py < HALO + RH && px >= HALO && px < HALO + TW)
          *reinterpret_cast<V*>(fdy + ((size_t)(sy0 + py - HALO) * p.W + sx0 + px - HALO) * CIN + st_slot * EP) = o;
      }
    }
  };
  if constexpr (FB) stage_patch_fold();
  else stage_patch(0);
  if constexpr (!DMA) store_w(0);
  stamp(p, 1);
  __syncthreads();     // (drains the two DMA stages in flight as well)
  stamp(p, 2);

  f32x16 acc[MF == 32 ? MR : 1][MF == 32 ? NR : 1];
  f32x4 acc4[MF == 16 ? MR4 : 1][MF == 16 ? NR4 : 1];
  if constexpr (MF == 32) {
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
      for (int j = 0; j < NR; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  } else {
#pragma unroll
    for (int i = 0; i < MR4; ++i)
#pragma unroll
      for (int j = 0; j < NR4; ++j) acc4[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
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

  const int r = lane & 31, h = lane >> 5;
  constexpr int ITERS = NTY * NTX * KCH;
  const char* aBase[MR];
#pragma unroll
  for (int i = 0; i < MR; ++i) {
    const int R = (wm * MR + i) * 32 + r;          // tile pixel of this lane's fragment row
    const int fs = R / SUBPIX, rr = R - fs * SUBPIX;
    aBase[i] = patch + (fs * PP + (rr / TW) * PW + rr % TW) * PITCH + h * 16;
  }
  int bOff[NR][2];
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    bOff[j][0] = lds_off(wn * NR * 32 + j * 32 + r, h);
    bOff[j][1] = lds_off(wn * NR * 32 + j * 32 + r, 2 + h);
  }
  // 16x16x32 operands: lane = row (pixel / output channel) % 16 + 16 * (K / 8): one fragment covers the whole 32-channel K chunk
  const char* aBase4[MF == 16 ? MR4 : 1];
  int bOff4[MF == 16 ? NR4 : 1];
  if constexpr (MF == 16) {
    const int r16 = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int i = 0; i < MR4; ++i) {
      const int R = wm * MR * 32 + i * 16 + r16;
      const int fs = R / SUBPIX, rr = R - fs * SUBPIX;
      aBase4[i] = patch + (fs * PP + (rr / TW) * PW + rr % TW) * PITCH + kg * 16;
    }
#pragma unroll
    for (int j = 0; j < NR4; ++j) bOff4[j] = lds_off(wn * NR * 32 + j * 16 + r16, kg);
  }
  // fully unrolled over (ky, kx, K chunk): every LDS offset, ring stage and wait count is a compile-time constant
  if constexpr (DMA) {
    // Weight ring of three LDS stages filled by LDS-DMA, fragments one K step AHEAD in registers:
    //   step `it` = [wait: stage it+1 landed (own share), all own LDS reads retired] -> barrier -> DMA for it+3 into the stage
    //   whose fragments (step it) were read during step it-1 -> issue the 8 fragment reads of step it+1 -> 8 MFMAs of step it
    //   on registers.
    // So no MFMA ever waits on an LDS read issued in its own step (two waves per SIMD cannot hide that latency), a DMA has two
    // steps to land, and a stage is overwritten only after a barrier that every wave reached with lgkmcnt(0): a ds_read that
    // is merely ISSUED can still be queued in the LDS pipe when another wave's DMA lands on the same bytes (seen with the
    // earlier read-in-step form as one wrong output channel of half a tile once per ~10^4 launches, tools/determinism*.py).
    V fa[2][2][MR], fb[2][2][NR];
    auto frag_load = [&](int it, V (&a)[2][MR], V (&bq)[2][NR]) {
      const int tap = it / KCH, kch = it - tap * KCH;
      int ky = tap / NTX, kx = tap - ky * NTX;     // window offset of the tap inside the patch
      if constexpr (S2) { ky = s2_tap(tap) / 3 == 0 ? 0 : 1; kx = s2_tap(tap) % 3 == 0 ? 0 : 1; }     // (tap 0 of an axis reads class pixel -1)
      const char* lB = wbuf + (GEO == 1 ? it : it % 3) * WBUF_BYTES;
      if constexpr (MF == 16) {       // (the [2][MR] / [2][NR] arrays hold the MR4 / NR4 fragments: same registers)
#pragma unroll
        for (int i = 0; i < MR4; ++i) a[i / MR][i % MR] = *reinterpret_cast<const V*>(aBase4[i] + (ky * PW + kx) * PITCH + kch * 64);
#pragma unroll
        for (int j = 0; j < NR4; ++j) bq[j / NR][j % NR] = *reinterpret_cast<const V*>(lB + bOff4[j]);
        return;
      }
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
#pragma unroll
        for (int i = 0; i < MR; ++i)
          a[ss][i] = *reinterpret_cast<const V*>(aBase[i] + (ky * PW + kx) * PITCH + kch * 64 + ss * 32);
#pragma unroll
        for (int j = 0; j < NR; ++j) bq[ss][j] = *reinterpret_cast<const V*>(lB + bOff[j][ss]);
      }
    };
    if constexpr (GEO != 1) { if (2 < ITERS) dma_w(wstage(2), 2); }
    frag_load(0, fa[0], fb[0]);
    // one K step; `it` is a constant once the loops below are unrolled (ring stage, wait count, tap offset)
    auto kstep = [&](const int it, const bool prefetch) __attribute__((always_inline)) {
      // s_waitcnt vmcnt(N) lgkmcnt(0) as the BUILTIN (simm16: vmcnt[3:0], expcnt[6:4] = 7 (no wait), lgkmcnt[11:8]): the
      // compiler's own wait-count pass sees it and does not add an lgkmcnt(0) in front of this step's MFMAs, which would
      // wait for the reads just issued for the NEXT step (it cannot see through an inline-asm wait).
      if constexpr (GEO != 1) {
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        if (it + 2 < ITERS) {
          if constexpr (NBW == 2) __builtin_amdgcn_s_waitcnt(0x0072);
          else __builtin_amdgcn_s_waitcnt(0x0071);
        } else {
          __builtin_amdgcn_s_waitcnt(0x0070);
        }
        __builtin_amdgcn_s_barrier();
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        if (it + 3 < ITERS) dma_w(wstage(it + 3), it % 3);
      }
      // (S2: no prefetch across a class boundary -- the next class's patch is staged after this step)
      if (prefetch) frag_load(it + 1, fa[(it + 1) & 1], fb[(it + 1) & 1]);
      // pin the order: left alone, the scheduler sinks these reads to just before their first use (shortest live range),
      // i.e. back into the next step, and hoists that step's MFMAs above the barrier -- the read-then-wait form again.
      // bf16: the first MFMA goes out right behind the barrier and the DMA / fragment reads are issued in the shadow of the
      // MFMAs (sched_group_barrier pipeline below); issuing all ten memory instructions first left the matrix pipe idle for
      // ~150 cycles per step.
      if constexpr (sizeof(T) != 2) __builtin_amdgcn_sched_barrier(0);
      if constexpr (MF == 16) {
#pragma unroll
        for (int i = 0; i < MR4; ++i)
#pragma unroll
          for (int j = 0; j < NR4; ++j)
            acc4[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[it & 1][i / MR][i % MR], fb[it & 1][j / NR][j % NR], acc4[i][j], 0, 0, 0);
      } else
#pragma unroll
      for (int ss = 0; ss < 2; ++ss)
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
          for (int j = 0; j < NR; ++j) {
            if constexpr (sizeof(T) == 2) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[it & 1][ss][i], fb[it & 1][ss][j], acc[i][j], 0, 0, 0);
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[it & 1][ss][i][e], fb[it & 1][ss][j][e], acc[i][j], 0, 0, 0);
            }
          }
      if constexpr (sizeof(T) == 2) {
        constexpr int NMFMA = MF == 16 ? MR4 * NR4 : 2 * MR * NR, NREAD = 2 * (MR + NR);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                     // MFMA
        if (GEO != 1 && it + 3 < ITERS) __builtin_amdgcn_sched_group_barrier(0x010, NBW, 0);   // the LDS-DMA of stage it+3
        if (prefetch) {
#pragma unroll
          for (int k = 1; k < NMFMA; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x100, (NREAD + NMFMA - 2) / (NMFMA - 1), 0);   // DS reads (as many groups as it takes)
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (kTwoLevel) {
        if ((it & 7) == 7 || it + 1 == ITERS) {
#pragma unroll
          for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int j = 0; j < NR; ++j) { acc2[i][j] += acc[i][j]; acc[i][j] = f32x16{}; }
        }
      }
    };
    if constexpr (!S2) {
#pragma unroll
      for (int it = 0; it < ITERS; ++it) kstep(it, it + 1 < ITERS);
    } else {
      // four parity classes of 1, 2, 2 and 4 taps: run a class out of LDS, stage the next one (every wave passed the last step's
      // barrier with its fragment reads retired: nobody reads the old patch any more), barrier, first fragments, go on
      constexpr int B1 = KCH, B2 = 3 * KCH, B3 = 5 * KCH;
#pragma unroll
      for (int it = 0; it < B1; ++it) kstep(it, it + 1 < B1);
      stage_patch(1); __syncthreads(); frag_load(B1, fa[B1 & 1], fb[B1 & 1]);
#pragma unroll
      for (int it = B1; it < B2; ++it) kstep(it, it + 1 < B2);
      stage_patch(2); __syncthreads(); frag_load(B2, fa[B2 & 1], fb[B2 & 1]);
#pragma unroll
      for (int it = B2; it < B3; ++it) kstep(it, it + 1 < B3);
      stage_patch(3); __syncthreads(); frag_load(B3, fa[B3 & 1], fb[B3 & 1]);
#pragma unroll
      for (int it = B3; it < ITERS; ++it) kstep(it, it + 1 < ITERS);
    }
    __syncthreads();
  } else {
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int tap = it / KCH, kch = it - tap * KCH;
      const int ky = tap / NTX, kx = tap - ky * NTX;     // window offset of the tap inside the patch
      const int buf = it & 1;
      if (it + 1 < ITERS) load_w(wstage(it + 1));
      const char* lB = wbuf + buf * WBUF_BYTES;
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        V a[MR], bb[NR];
#pragma unroll
        for (int i = 0; i < MR; ++i)
          a[i] = *reinterpret_cast<const V*>(aBase[i] + (ky * PW + kx) * PITCH + kch * 64 + ss * 32);
#pragma unroll
        for (int j = 0; j < NR; ++j) bb[j] = *reinterpret_cast<const V*>(lB + bOff[j][ss]);
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
          for (int j = 0; j < NR; ++j) {
            if constexpr (sizeof(T) == 2) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], bb[j], acc[i][j], 0, 0, 0);
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], bb[j][e], acc[i][j], 0, 0, 0);
            }
          }
      }
      if constexpr (kTwoLevel) {
        if ((it & 7) == 7 || it + 1 == ITERS) {
#pragma unroll
          for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int j = 0; j < NR; ++j) { acc2[i][j] += acc[i][j]; acc[i][j] = f32x16{}; }
        }
      }
      if (it + 1 < ITERS) store_w(buf ^ 1);
      __syncthreads();
    }
  }
  if constexpr (kTwoLevel) {
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
      for (int j = 0; j < NR; ++j) acc[i][j] = acc2[i][j];
  }

  // NHWC pixel index of tile pixel ml in the output (TW == 32 forms)
  auto out_m = [&](int ml) -> size_t {
    if constexpr (TR) return (size_t)b * 4 * HW + (size_t)(2 * (ty0 + ml / TW) + CPY) * (2 * p.W) + 2 * (tx0 + ml % TW) + CPX;
    else return (size_t)b * OHW + (size_t)(ty0 + ml / TW) * OW + tx0 + ml % TW;
  };
  stamp(p, 3);
  // ---- epilogue of the 16x16x32 form (128 x 128 tile, bf16 NHWC output, no residual): ONE pass through a bf16 image.
  // A lane holds 4 consecutive pixels of one channel per 16x16 tile, so it writes them (+ bias, rounded) as ONE 8-byte store into a
  // CHANNEL-major image [128 channels][128 pixels] whose 16-byte chunks are XOR-swizzled (cdna_hip_programming.md T10, image (b));
  // ds_read_b64_tr_b16 hands them back pixel-major: lane i of a 16-lane group gets pixel 16 it + i, 4 channels per read, 2 reads =
  // the 16-byte NHWC vector it stores.  32 LDS instructions and 64 KB of LDS traffic per thread-tile instead of 80 and 128 KB for
  // the two fp32 passes below, two barriers instead of four; same values (fp32 accumulator + bias, one rounding), same statistics
  // up to the order of their fixed-order sums.
  if constexpr (MF == 16 && TW == 32 && (GEO == 0 || TR || S2) && MR4 == 4 && NR4 == 4) {
    if (p.y && !p.y_nchw && !p.residual && p.Cout % BN == 0 && p.epi16) {
      typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_e;
      char* E16 = smem;
      auto eoff = [](int ch, int chunk) { return 256 * ch + 16 * (chunk ^ (((ch & 3) << 2) | ((ch >> 2) & 3))); };
      {
        const int c15 = lane & 15, rg = lane >> 4;
#pragma unroll
        for (int j = 0; j < NR4; ++j) {
          const int col = wn * NR * 32 + j * 16 + c15;
          const float bv = p.bias ? p.bias[n0 + col] : 0.f;
#pragma unroll
          for (int i = 0; i < MR4; ++i) {
            const int r0 = wm * MR * 32 + i * 16 + 4 * rg;
            bf16x4_e v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (bf16_t)(acc4[i][j][e] + bv);
            *reinterpret_cast<bf16x4_e*>(E16 + eoff(col, r0 >> 3) + 8 * ((r0 >> 2) & 1)) = v;
          }
        }
      }
      __syncthreads();
      const int slot = 4 * wid + (lane >> 4), cb = slot * EP, n = n0 + cb;     // this thread's 8 channels, all its vectors
      const int li = lane & 15, q4 = li >> 2, p4 = li & 3;
      EpiStats<T> est;
      est.init(p, b, n);
      if (est.kind == 1) {      // shift of the forward statistics: the tile's first pixel (uniform over the threads of a slot)
#pragma unroll
        for (int e = 0; e < EP; ++e) est.a0[e] = (float)*reinterpret_cast<const bf16_t*>(E16 + eoff(cb + e, 0));
      }
      const T* __restrict__ nby = est.kind == 2 ? reinterpret_cast<const T*>(p.nb_y) : reinterpret_cast<const T*>(p.w);
      T* __restrict__ y = reinterpret_cast<T*>(p.y);
      V ypre[8];
      size_t mrow[8];
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        mrow[it] = out_m(16 * it + li);
        ypre[it] = *reinterpret_cast<const V*>(nby + (est.kind == 2 ? mrow[it] * p.Cout + n : 0));
      }
      typedef __attribute__((address_space(3))) bf16x4_e* lptr;
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int px = 16 * it + 4 * p4;                     // first of the 4 pixels whose address this lane supplies
        const int ck = px >> 3, hf = 8 * ((px >> 2) & 1);
        const bf16x4_e lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lptr)(E16 + eoff(cb + q4, ck) + hf));
        const bf16x4_e hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lptr)(E16 + eoff(cb + 4 + q4, ck) + hf));
        V o;
        o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3]; o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
        *reinterpret_cast<V*>(y + mrow[it] * p.Cout + n) = o;
        est.add_pre(p, o, ypre[it]);
      }
      if (est.kind) {           // the 16 lanes of a group share the slot: fixed-order butterfly, lane 0 of the group writes
#pragma unroll
        for (int e = 0; e < EP; ++e) {      // (lane ^ 1, 2, 4, 8 by DPP, pwr_common.h: the __shfl_xor loop's sums without its 64 ds_bpermute_b32)
          est.s1[e] = lane_xor_add<1>(est.s1[e]); est.s2[e] = lane_xor_add<1>(est.s2[e]);
        }
#pragma unroll
        for (int e = 0; e < EP; ++e) { est.s1[e] = lane_xor_add<2>(est.s1[e]); est.s2[e] = lane_xor_add<2>(est.s2[e]); }
#pragma unroll
        for (int e = 0; e < EP; ++e) { est.s1[e] = lane_xor_add<4>(est.s1[e]); est.s2[e] = lane_xor_add<4>(est.s2[e]); }
#pragma unroll
        for (int e = 0; e < EP; ++e) { est.s1[e] = lane_xor_add<8>(est.s1[e]); est.s2[e] = lane_xor_add<8>(est.s2[e]); }
        if (li == 0) {
          const size_t srow = (size_t)b * (p.st_nchunks ? p.st_nchunks : tiles_img) + p.st_chunk0 + tr;
          float* out = est.kind == 1 ? p.st_partial + (srow * 3) * p.Cout : p.nb_partial + (srow * 2) * p.Cout;
#pragma unroll
          for (int e = 0; e < EP; ++e) {
            out[n + e] = est.s1[e];
            out[(size_t)p.Cout + n + e] = est.s2[e];
            if (est.kind == 1) out[(size_t)2 * p.Cout + n + e] = est.a0[e];
          }
        }
      }
      stamp(p, 4);
      return;
    }
  }
  // ---- epilogue: accumulators -> LDS (fp32, 64 tile pixels at a time) -> coalesced 16-byte stores
  float* E = reinterpret_cast<float*>(smem);
  constexpr int PASSES = BM / EROWS;
  constexpr int CPRS = BN / EP;
  EpiStats<T> est;
  if constexpr (TW == 32) est.init(p, b, n0 + (tid % CPRS) * EP);
  // TW == 32: everything the copy loop needs from global memory is fetched NOW, branch-free (a load inside a branch makes hipcc
  // wait vmcnt(0) at the join; unused streams read one dummy vector), and arrives while the accumulators go to LDS: bias, the
  // residual vectors, and -- for the norm-backward sums -- the forward activations.  Fetched inside the loop, each of its 4
  // iterations per pass exposed a global round trip (the nb form cost 11 us more than the plain one).
  constexpr int NCI = (EROWS * CPRS + NT - 1) / NT;        // copy-loop iterations per pass
  static_assert(TW != 32 || (EROWS * CPRS) % NT == 0, "whole iterations");
  V ypre[NCI], rpre[NCI];                                   // (per pass: both passes at once cost the second workgroup per CU)
  float bias_r[EP];
  const T* __restrict__ nby = est.kind == 2 ? reinterpret_cast<const T*>(p.nb_y) : reinterpret_cast<const T*>(p.w);
  const T* __restrict__ resp = p.residual ? reinterpret_cast<const T*>(p.residual) : reinterpret_cast<const T*>(p.w);
  if constexpr (TW == 32) {
    const int n = n0 + (tid % CPRS) * EP;
    const float* bp = p.bias ? p.bias + (n < p.Cout ? n : 0) : reinterpret_cast<const float*>(p.w);
#pragma unroll
    for (int e = 0; e < EP; ++e) { const float bv = bp[e]; bias_r[e] = p.bias ? bv : 0.f; }
  }
#pragma unroll
  for (int ps = 0; ps < PASSES; ++ps) {
    if constexpr (TW == 32) {
      const int n = n0 + (tid % CPRS) * EP;
      const bool nok = n < p.Cout;
#pragma unroll
      for (int it = 0; it < NCI; ++it) {
        const int ml = ps * EROWS + (tid + it * NT) / CPRS;
        const size_t m = out_m(ml);
        const size_t idx = nok ? m * p.Cout + n : 0;
        ypre[it] = *reinterpret_cast<const V*>(nby + (est.kind == 2 ? idx : 0));
        rpre[it] = *reinterpret_cast<const V*>(resp + (p.residual ? idx : 0));
      }
    }
    const int wrow0 = wm * MR * 32;
    if (wrow0 / EROWS == ps) {
      const int er0 = wrow0 - ps * EROWS;
      if constexpr (MF == 16) {       // D[i][j] of a 16x16 tile: lane = j + 16 * (i / 4), element i % 4
#pragma unroll
        for (int i = 0; i < MR4; ++i)
#pragma unroll
          for (int j = 0; j < NR4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              E[(er0 + i * 16 + 4 * (lane >> 4) + e) * EPITCH + wn * NR * 32 + j * 16 + (lane & 15)] = acc4[i][j][e];
      } else
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
    if constexpr (TW == 32) {
      if (ps == 0 && est.kind == 1) {
        const int n = n0 + (tid % CPRS) * EP;
        if (n < p.Cout) {
#pragma unroll
          for (int e = 0; e < EP; ++e) est.a0[e] = E[(tid % CPRS) * EP + e] + bias_r[e];     // the shift (set_shift with the bias in registers)
        }
      }
    }
    if (p.y) {
      T* __restrict__ y = reinterpret_cast<T*>(p.y);
      const T* __restrict__ res = reinterpret_cast<const T*>(p.residual);
      constexpr int CPR = BN / EP;
      if constexpr (TW == 32) {
#pragma unroll
        for (int it = 0; it < NCI; ++it) {
          const int c = tid + it * NT;
          const int row = c / CPR, cc = (c - row * CPR) * EP;
          const int ml = ps * EROWS + row, n = n0 + cc;
          const size_t m = out_m(ml);
          if (n < p.Cout) {
            float v[EP];
#pragma unroll
            for (int e = 0; e < EP; ++e) v[e] = E[row * EPITCH + cc + e] + bias_r[e];
            if (res) {
#pragma unroll
              for (int e = 0; e < EP; ++e) v[e] += Elem<T>::to_f(rpre[it][e]);
            }
            V o;
#pragma unroll
            for (int e = 0; e < EP; ++e) o[e] = Elem<T>::from_f(v[e]);
            *reinterpret_cast<V*>(y + m * p.Cout + n) = o;
            est.add_pre(p, o, ypre[it]);
          }
        }
      } else {
      for (int c = tid; c < EROWS * CPR; c += NT) {
        const int row = c / CPR, cc = (c - row * CPR) * EP;
        const int ml = ps * EROWS + row, n = n0 + cc;
        size_t m;
        bool mvalid = true;
        m = (size_t)(L0 + ml); mvalid = L0 + ml < Mtot;
        if (n < p.Cout && mvalid) {
          float v[EP];
#pragma unroll
          for (int e = 0; e < EP; ++e) v[e] = E[row * EPITCH + cc + e];
          if (p.bias) {
#pragma unroll
            for (int e = 0; e < EP; ++e) v[e] += p.bias[n + e];
          }
          if (res) {
            V rv = *reinterpret_cast<const V*>(res + m * p.Cout + n);
#pragma unroll
            for (int e = 0; e < EP; ++e) v[e] += Elem<T>::to_f(rv[e]);
          }
          V o;
#pragma unroll
          for (int e = 0; e < EP; ++e) o[e] = Elem<T>::from_f(v[e]);
          *reinterpret_cast<V*>(y + m * p.Cout + n) = o;
        }
      }
      }
    }
    if (p.y_nchw) {
      for (int c = tid; c < EROWS * BN; c += NT) {
        const int col = c / EROWS, row = c - col * EROWS;
        const int ml = ps * EROWS + row, n = n0 + col;
        int ob = b, opix;
        bool mvalid = true;
        if constexpr (TW == 32) opix = (ty0 + ml / TW) * OW + tx0 + ml % TW;
        else { mvalid = L0 + ml < Mtot; ob = (int)((L0 + ml) / HW); opix = (int)(L0 + ml - (long long)ob * HW); }
        if (n < p.Cout && mvalid) {
          float v = E[row * EPITCH + col];
          if (p.bias) v += p.bias[n];
          p.y_nchw[((size_t)ob * p.Cout + n) * (TW == 32 ? OHW : HW) + opix] = v;
        }
      }
    }
    __syncthreads();
  }
  if constexpr (TW == 32) est.template finish<CPRS, NT>(p, E, b, p.st_chunk0 + tr, p.st_nchunks ? p.st_nchunks : tiles_img, n0);
  stamp(p, 4);
}

template <typename T, int CIN, int WM, int WN, int MR, int NR, bool DMA, int TW = 32, int GEO = 0, int MF = 32>
__global__ __launch_bounds__(WM * WN * 64, (PWR_OCC_HINT && sizeof(T) == 2 && WM * WN == 4) ? 2 : 1) void conv3x3_patch_kernel(ConvParams p) {
  conv3x3_patch_body<T, CIN, WM, WN, MR, NR, DMA, TW, GEO, MF>(p);
}

// Two convs of the same shape and kernel variant in ONE launch (blockIdx.z picks the job): the two regression heads of a stage run the
// same three 128 -> 128 convs on different tensors (model.py:54-65 / :103-114), and a launch boundary between two full-chip launches of
// this kernel costs 8 - 9 us (tools/launch_bubble.py: 2 x B = 32 takes 85.5 us, 1 x B = 64 takes 76.4 us).
struct ConvPair { ConvParams a, b; };
template <typename T, int CIN, int WM, int WN, int MR, int NR, bool DMA, int TW = 32, int GEO = 0, int MF = 32, bool FB = false>
__global__ __launch_bounds__(WM * WN * 64, (PWR_OCC_HINT && sizeof(T) == 2 && WM * WN == 4) ? 2 : 1) void conv3x3_patch_pair_kernel(ConvPair g) {
  conv3x3_patch_body<T, CIN, WM, WN, MR, NR, DMA, TW, GEO, MF, false, FB>(blockIdx.z ? g.b : g.a);
}

// The four parity classes of a stride-2 data gradient (GEO 2 .. 5) in ONE launch: blockIdx.z = 0 .. 3 runs the class with 4, 2, 2, 1 taps --
// the order the four launches had -- into its own rows of the norm-backward slab.  Same code per class, so the same bits; three launch
// boundaries and three ragged launch ends less, and the one-tap class's workgroups (memory-bound) run beside the four-tap class's.
template <typename T, int WM, int WN, int MR, int NR, int MF>
__global__ __launch_bounds__(WM * WN * 64, (PWR_OCC_HINT && WM * WN == 4) ? 2 : 1) void conv3x3_patch_tr4_kernel(ConvParams p) {
  constexpr int L5 = patch_lds_bytes<T, 128, WM, WN, MR, NR, true, 32, 5>();      // (the four-tap class has the largest patch)
  static_assert(L5 >= patch_lds_bytes<T, 128, WM, WN, MR, NR, true, 32, 4>() && L5 >= patch_lds_bytes<T, 128, WM, WN, MR, NR, true, 32, 3>() &&
                L5 >= patch_lds_bytes<T, 128, WM, WN, MR, NR, true, 32, 2>(), "class (1, 1) has the largest LDS footprint");
  __shared__ __attribute__((aligned(16))) char smem[L5];
  const int tiles = (p.H / 4) * (p.W / 32);
  switch (blockIdx.z) {
    case 0: p.st_chunk0 = 3 * tiles; conv3x3_patch_body<T, 128, WM, WN, MR, NR, true, 32, 5, MF, true>(p, smem); break;
    case 1: p.st_chunk0 = 2 * tiles; conv3x3_patch_body<T, 128, WM, WN, MR, NR, true, 32, 4, MF, true>(p, smem); break;
    case 2: p.st_chunk0 = 1 * tiles; conv3x3_patch_body<T, 128, WM, WN, MR, NR, true, 32, 3, MF, true>(p, smem); break;
    default: p.st_chunk0 = 0; conv3x3_patch_body<T, 128, WM, WN, MR, NR, true, 32, 2, MF, true>(p, smem); break;
  }
}

// small square maps of the inner hourglass levels (64 -> 64 channels): whole images per tile
static bool small_map(const ConvParams& p, int dtype) {
  static const bool on = (PWR_DBG_ENV("PWR_PATCH_SMALL", 1) != 0);
  if (!on || p.H != p.W || p.Cin != 64 || pick_bn(p.Cout) != 64) return false;
  if (p.W == 2) return dtype == PWR_BF16;     // fp32: 16 channel slots > the 8 threads a 2x2 image gets
  return p.W == 4 || p.W == 8 || p.W == 16;
}

static bool conv1x1_applicable(const ConvParams& p, int dtype) {
  static const bool on = (PWR_DBG_ENV("PWR_PATCH_1X1", 1) != 0);
  return on && dtype == PWR_BF16 && p.mode == 0 && p.ksize == 1 && p.stride == 1 && p.pad == 0 && p.W % 32 == 0 && p.H % 4 == 0 &&
         (p.Cin == 32 || p.Cin == 64 || p.Cin == 128) && p.y != nullptr;
}

// forward of a stride-2 3x3 conv (GEO 6): bf16, tiles of 4 x 32 OUTPUT pixels
static bool conv_s2_applicable(const ConvParams& p, int dtype) {
  static const bool on = (PWR_DBG_ENV("PWR_PATCH_S2", 1) != 0);
  return on && dtype == PWR_BF16 && p.mode == 0 && p.ksize == 3 && p.stride == 2 && p.pad == 1 && p.H % 2 == 0 && p.W % 2 == 0 &&
         p.Wo % 32 == 0 && p.Ho % 4 == 0 && (p.Cin == 32 || p.Cin == 64 || p.Cin == 128) && p.y != nullptr && !p.y_nchw;
}

bool conv_patch_applicable(const ConvParams& p, int dtype) {
  if (conv1x1_applicable(p, dtype)) return true;
  if (conv_s2_applicable(p, dtype)) return true;
  if (!(p.mode == 0 && p.ksize == 3 && p.stride == 1 && p.pad == 1)) return false;
  if (small_map(p, dtype)) return true;
  return p.W % 32 == 0 && p.H % 4 == 0 && (p.Cin == 32 || p.Cin == 64 || p.Cin == 128);
}

// 128 -> (33..64) channels on big maps (the stem's 64 <- 128 data gradient at 128x128): 8 x 32-pixel tiles, 8 waves, ONE workgroup per
// CU -- with 64 output channels a 4 x 32 tile does half the MFMA work per staged patch, the 8 x 32 tile stages 1.33x instead of 1.6x
// the input and streams the weights once per 256 pixels
static bool big64(const ConvParams& p, int dtype) {
  static const bool on = (PWR_DBG_ENV("PWR_PATCH_BIG64", 0) != 0);   // measured: 103 vs 97 us isolated, 6.95 vs 6.90 ms/step -> off
  return on && dtype == PWR_BF16 && p.mode == 0 && p.ksize == 3 && p.stride == 1 && p.pad == 1 && p.Cin == 128 && pick_bn(p.Cout) == 64 &&
         p.H % 8 == 0 && p.W % 32 == 0 && p.H * p.W >= 128 * 128;
}

int conv_patch_stats_chunks(const ConvParams& p, int dtype) {
  if (conv1x1_applicable(p, dtype)) return (p.H / 4) * (p.W / 32);
  if (conv_s2_applicable(p, dtype)) return (p.Ho / 4) * (p.Wo / 32);
  if (big64(p, dtype)) return (p.H / 8) * (p.W / 32);
  if (!conv_patch_applicable(p, dtype) || small_map(p, dtype)) return 0;
  return (p.H / 4) * (p.W / 32);
}

template <typename T>
static int launch_patch_small(const ConvParams& p, hipStream_t s) {
  constexpr bool dma = sizeof(T) == 2;
  dim3 grid((unsigned)(((long long)p.B * p.H * p.W + 127) / 128), p.CoutPad / 64), block(256);
  if (p.W == 16) hipLaunchKernelGGL((conv3x3_patch_kernel<T, 64, 2, 2, 2, 1, dma, 16>), grid, block, 0, s, p);
  else if (p.W == 8) hipLaunchKernelGGL((conv3x3_patch_kernel<T, 64, 2, 2, 2, 1, dma, 8>), grid, block, 0, s, p);
  else if (p.W == 4) hipLaunchKernelGGL((conv3x3_patch_kernel<T, 64, 2, 2, 2, 1, dma, 4>), grid, block, 0, s, p);
  else {
    if constexpr (sizeof(T) == 2) hipLaunchKernelGGL((conv3x3_patch_kernel<T, 64, 2, 2, 2, 1, dma, 2>), grid, block, 0, s, p);
    else return (int)hipErrorInvalidValue;
  }
  return (int)hipGetLastError();
}

template <typename T, int CIN>
static int launch_patch_cin(const ConvParams& p, hipStream_t s) {
  const int bn = pick_bn(p.Cout);
  dim3 grid(p.B * (p.H / 4) * (p.W / 32), p.CoutPad / bn), block(256);
  static const bool dma = (PWR_DBG_ENV("PWR_PATCH_DMA", 1) != 0);
  const_cast<ConvParams&>(p).stamps = g_stamps; const_cast<ConvParams&>(p).dbg_delay = g_delay;
  // fp32 (parity mode) patches leave no room for a third weight stage next to a second workgroup: register staging
  if (dma && sizeof(T) == 2) {
    static const bool big = (PWR_DBG_ENV("PWR_PATCH_BIG", 0) != 0);   // measured: 61 us vs 59 us for the 4x32 tile -> off
    if constexpr (sizeof(T) == 2 && CIN == 128) {
      if (bn == 128 && p.H % 8 == 0 && big && !p.st_partial && !p.nb_partial) {
        // 8 waves, 8x32-pixel tile: the per-CU weight stream from L2 (the limiter of the 4x32 form) is halved
        dim3 g8(p.B * (p.H / 8) * (p.W / 32), p.CoutPad / bn);
        hipLaunchKernelGGL((conv3x3_patch_kernel<T, CIN, 4, 2, 2, 2, true>), g8, dim3(512), 0, s, p);
        return (int)hipGetLastError();
      }
    }
    if constexpr (sizeof(T) == 2 && CIN == 128) {
      if (big64(p, PWR_BF16)) {
        dim3 g8(p.B * (p.H / 8) * (p.W / 32), p.CoutPad / bn);
        hipLaunchKernelGGL((conv3x3_patch_kernel<T, CIN, 4, 2, 2, 1, true>), g8, dim3(512), 0, s, p);
        return (int)hipGetLastError();
      }
    }
    // the 128 -> 128 tile (the heads' convs and their data gradients: 72 % of the FLOPs) issues 16x16x32 MFMAs: bit-identical
    // results, 2 - 9 % less time isolated, 0.04 ms per train step (PWR_PATCH_MF16=0 in the debug build: the 32x32x16 form; extending
    // it to the other tiles, the 1x1 form and the stride-2 classes gave nothing more: profiles/r3_experiments.md section 8)
    if constexpr (sizeof(T) == 2) {
      const int mf16 = PWR_DBG_ENV("PWR_PATCH_MF16", 1);      // 0: 32x32x16 everywhere, 1: 16x16x32 for 128 -> 128 only, 2: for every Cin at 128 output channels (measured: train step +1.3 %, inference -1 %)
      if (bn == 128 && (mf16 == 2 || (mf16 == 1 && CIN == 128))) {
        const_cast<ConvParams&>(p).epi16 = PWR_DBG_ENV("PWR_PATCH_EPI16", 1);
        hipLaunchKernelGGL((conv3x3_patch_kernel<T, CIN, 2, 2, 2, 2, true, 32, 0, 16>), grid, block, 0, s, p);
        return (int)hipGetLastError();
      }
    }
    if (bn == 128) hipLaunchKernelGGL((conv3x3_patch_kernel<T, CIN, 2, 2, 2, 2, true>), grid, block, 0, s, p);
    else if (bn == 64) hipLaunchKernelGGL((conv3x3_patch_kernel<T, CIN, 2, 2, 2, 1, true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((conv3x3_patch_kernel<T, CIN, 4, 1, 1, 1, true>), grid, block, 0, s, p);
  } else {
    if (bn == 128) hipLaunchKernelGGL((conv3x3_patch_kernel<T, CIN, 2, 2, 2, 2, false>), grid, block, 0, s, p);
    else if (bn == 64) hipLaunchKernelGGL((conv3x3_patch_kernel<T, CIN, 2, 2, 2, 1, false>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((conv3x3_patch_kernel<T, CIN, 4, 1, 1, 1, false>), grid, block, 0, s, p);
  }
  return (int)hipGetLastError();
}

template <typename T>
static int launch_patch_t(const ConvParams& p, hipStream_t s) {
  if (p.Cin == 128) return launch_patch_cin<T, 128>(p, s);
  if (p.Cin == 64) return launch_patch_cin<T, 64>(p, s);
  return launch_patch_cin<T, 32>(p, s);
}

#ifdef PWR_DEBUG_BUILD
long long* wstat_stamps() { return g_stamps; }
void set_debug_stamps(long long* ptr) { g_stamps = ptr; }
void set_debug_delay(int d) { g_delay = d; }
#endif

template <int CIN>
static int launch_patch_1x1(const ConvParams& p, hipStream_t s) {
  const int bn = pick_bn(p.Cout);
  dim3 grid(p.B * (p.H / 4) * (p.W / 32), p.CoutPad / bn), block(256);
  if (bn == 128) hipLaunchKernelGGL((conv3x3_patch_kernel<bf16_t, CIN, 2, 2, 2, 2, true, 32, 1>), grid, block, 0, s, p);
  else if (bn == 64) hipLaunchKernelGGL((conv3x3_patch_kernel<bf16_t, CIN, 2, 2, 2, 1, true, 32, 1>), grid, block, 0, s, p);
  else hipLaunchKernelGGL((conv3x3_patch_kernel<bf16_t, CIN, 4, 1, 1, 1, true, 32, 1>), grid, block, 0, s, p);
  return (int)hipGetLastError();
}

// data gradient of a stride-2 3x3 conv (ConvParams::mode 1: x = dy [B,H,W,Cin], output [B,2H,2W,Cout]) as four parity classes -- ONE launch
// (conv3x3_patch_tr4_kernel, round 6; PWR_TR2_ONE=0 in the debug build: the four launches of rounds 3 - 5, same bits)
int conv_tr2_stats_chunks(const ConvParams& p, int dtype);
bool conv_tr2_applicable(const ConvParams& p, int dtype) {
  static const bool on = (PWR_DBG_ENV("PWR_PATCH_TR2", 1) != 0);
  return on && dtype == PWR_BF16 && p.mode == 1 && p.ksize == 3 && p.Cin == 128 && p.W % 32 == 0 && p.H % 4 == 0 && p.y != nullptr &&
         !p.y_nchw && !p.st_partial && !p.in_norm && (!p.nb_partial || conv_tr2_stats_chunks(p, dtype) > 0);
}
// norm-backward sums from the four class launches' epilogues (the 128-channel one-pass epilogue): slab rows per sample, 0 = not offered
int conv_tr2_stats_chunks(const ConvParams& p, int dtype) {
  if (dtype != PWR_BF16 || p.mode != 1 || p.ksize != 3 || p.Cin != 128 || p.W % 32 || p.H % 4 || p.Cout % 128 || p.residual) return 0;
  return 4 * (p.H / 4) * (p.W / 32);
}

template <int GEO>
static void launch_tr2_class(const ConvParams& p, hipStream_t s) {
  const int bn = pick_bn(p.Cout);
  dim3 grid(p.B * (p.H / 4) * (p.W / 32), p.CoutPad / bn), block(256);
  // 128 output channels: the 16x16x32 form with the one-pass bf16 epilogue -- a class has 1 - 4 taps only, so the epilogue is a large
  // part of its launch
  static const bool mf16 = PWR_DBG_ENV("PWR_TR2_MF16", 1) != 0;
  if (bn == 128 && mf16 && !p.residual && p.Cout % 128 == 0) {
    hipLaunchKernelGGL((conv3x3_patch_kernel<bf16_t, 128, 2, 2, 2, 2, true, 32, GEO, 16>), grid, block, 0, s, p);
    return;
  }
  if (bn == 128) hipLaunchKernelGGL((conv3x3_patch_kernel<bf16_t, 128, 2, 2, 2, 2, true, 32, GEO>), grid, block, 0, s, p);
  else if (bn == 64) hipLaunchKernelGGL((conv3x3_patch_kernel<bf16_t, 128, 2, 2, 2, 1, true, 32, GEO>), grid, block, 0, s, p);
  else hipLaunchKernelGGL((conv3x3_patch_kernel<bf16_t, 128, 4, 1, 1, 1, true, 32, GEO>), grid, block, 0, s, p);
}

int launch_conv_tr2(const ConvParams& p0, hipStream_t s) {
  ConvParams p = p0;
  p.stamps = nullptr;
  const int tiles = (p.H / 4) * (p.W / 32);
  p.st_nchunks = 4 * tiles;      // (norm-backward sums: one slab, class c at rows c * tiles ... of every sample)
  static const bool one = PWR_DBG_ENV("PWR_TR2_ONE", 1) != 0;
  if (one) {
    const int bn = pick_bn(p.Cout);
    dim3 grid(p.B * tiles, p.CoutPad / bn, 4), block(256);
    static const bool mf16 = PWR_DBG_ENV("PWR_TR2_MF16", 1) != 0;
    if (bn == 128 && mf16 && !p.residual && p.Cout % 128 == 0) hipLaunchKernelGGL((conv3x3_patch_tr4_kernel<bf16_t, 2, 2, 2, 2, 16>), grid, block, 0, s, p);
    else if (bn == 128) hipLaunchKernelGGL((conv3x3_patch_tr4_kernel<bf16_t, 2, 2, 2, 2, 32>), grid, block, 0, s, p);
    else if (bn == 64) hipLaunchKernelGGL((conv3x3_patch_tr4_kernel<bf16_t, 2, 2, 2, 1, 32>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((conv3x3_patch_tr4_kernel<bf16_t, 4, 1, 1, 1, 32>), grid, block, 0, s, p);
    return (int)hipGetLastError();
  }
  p.st_chunk0 = 3 * tiles; launch_tr2_class<5>(p, s);     // (the four-tap class first: the longest)
  p.st_chunk0 = 2 * tiles; launch_tr2_class<4>(p, s);
  p.st_chunk0 = 1 * tiles; launch_tr2_class<3>(p, s);
  p.st_chunk0 = 0; launch_tr2_class<2>(p, s);
  return (int)hipGetLastError();
}

template <int CIN>
static int launch_patch_s2(const ConvParams& p, hipStream_t s) {
  const int bn = pick_bn(p.Cout);
  dim3 grid(p.B * (p.Ho / 4) * (p.Wo / 32), p.CoutPad / bn), block(256);
  if constexpr (CIN == 128) {
    static const bool mf16 = PWR_DBG_ENV("PWR_S2_MF16", 1) != 0;       // (the 128 -> 128 tile: 16x16x32 MFMAs and the one-pass epilogue, as the heads' convs)
    if (bn == 128 && mf16) {
      const_cast<ConvParams&>(p).epi16 = 1;
      hipLaunchKernelGGL((conv3x3_patch_kernel<bf16_t, CIN, 2, 2, 2, 2, true, 32, 6, 16>), grid, block, 0, s, p);
      return (int)hipGetLastError();
    }
  }
  if (bn == 128) hipLaunchKernelGGL((conv3x3_patch_kernel<bf16_t, CIN, 2, 2, 2, 2, true, 32, 6>), grid, block, 0, s, p);
  else if (bn == 64) hipLaunchKernelGGL((conv3x3_patch_kernel<bf16_t, CIN, 2, 2, 2, 1, true, 32, 6>), grid, block, 0, s, p);
  else hipLaunchKernelGGL((conv3x3_patch_kernel<bf16_t, CIN, 4, 1, 1, 1, true, 32, 6>), grid, block, 0, s, p);
  return (int)hipGetLastError();
}

// The pair form exists for the kernel variants of the heads' convs and of their data gradients: bf16, 3x3 stride 1, 4 x 32-pixel tiles,
// 128 output channels, 32 / 64 / 128 input channels (128: 16x16x32 MFMAs with the one-pass epilogue, as the single launch picks it;
// 32 / 64 -- the heads' last data gradient, J padded to a K chunk -- the 32x32x16 form); both jobs of one geometry.
bool conv_patch_pair_applicable(const ConvParams& a, const ConvParams& b, int dtype) {
  static const bool on = (PWR_DBG_ENV("PWR_PATCH_PAIR", 1) != 0);
  auto ok = [&](const ConvParams& p) {
    return dtype == PWR_BF16 && p.mode == 0 && p.ksize == 3 && p.stride == 1 && p.pad == 1 && (p.Cin == 128 || p.Cin == 64 || p.Cin == 32) && p.Cout == 128 &&
           p.W % 32 == 0 && p.H % 4 == 0 && p.y != nullptr && !p.y_nchw && !p.residual && !small_map(p, dtype);
  };
  return on && ok(a) && ok(b) && a.B == b.B && a.H == b.H && a.W == b.W && a.Cin == b.Cin && PWR_DBG_ENV("PWR_PATCH_MF16", 1) == 1 && PWR_DBG_ENV("PWR_PATCH_BIG", 0) == 0;
}
int launch_conv_patch_pair(const ConvParams& a, const ConvParams& b, hipStream_t s) {
  if (!a.fb_y && !b.fb_y && conv_wstat_pair_applicable(a, b, PWR_BF16)) return launch_conv_wstat(a, &b, s);
  if (a.w_frag || b.w_frag) return PWR_EINVAL;       // a fragment-order pack and a launch that is not conv_wstat.hip's
  ConvPair g{a, b};
  g.a.epi16 = g.b.epi16 = 1;
  g.a.stamps = g.b.stamps = nullptr;
  g.a.dbg_delay = g.b.dbg_delay = PWR_DBG_ENV("PWR_PAIR_DELAY", 0);
  dim3 grid(a.B * (a.H / 4) * (a.W / 32), a.CoutPad / 128, 2), block(256);
  if ((a.fb_y != nullptr) != (b.fb_y != nullptr)) return PWR_EINVAL;
  if (a.fb_y) {        // the fold form exists for the heads' shape only
    if (a.Cin != 128 || !a.fb_state || !a.fb_partial || !a.fb_dy || !b.fb_state || !b.fb_partial || !b.fb_dy || a.fb_pchunks < 1 || a.in_norm || b.in_norm) return PWR_EUNSUPPORTED;
    hipLaunchKernelGGL((conv3x3_patch_pair_kernel<bf16_t, 128, 2, 2, 2, 2, true, 32, 0, 16, true>), grid, block, 0, s, g);
    return (int)hipGetLastError();
  }
  if (a.Cin == 128) hipLaunchKernelGGL((conv3x3_patch_pair_kernel<bf16_t, 128, 2, 2, 2, 2, true, 32, 0, 16>), grid, block, 0, s, g);
  else if (a.Cin == 64) hipLaunchKernelGGL((conv3x3_patch_pair_kernel<bf16_t, 64, 2, 2, 2, 2, true>), grid, block, 0, s, g);
  else hipLaunchKernelGGL((conv3x3_patch_pair_kernel<bf16_t, 32, 2, 2, 2, 2, true>), grid, block, 0, s, g);
  return (int)hipGetLastError();
}

int launch_conv_patch(const ConvParams& p, int dtype, hipStream_t s) {
  if (conv_s2_applicable(p, dtype)) {
    const_cast<ConvParams&>(p).stamps = g_stamps; const_cast<ConvParams&>(p).dbg_delay = g_delay;
    if (p.Cin == 128) return launch_patch_s2<128>(p, s);
    if (p.Cin == 64) return launch_patch_s2<64>(p, s);
    return launch_patch_s2<32>(p, s);
  }
  if (conv1x1_applicable(p, dtype)) {
    const_cast<ConvParams&>(p).stamps = g_stamps; const_cast<ConvParams&>(p).dbg_delay = g_delay;
    if (p.Cin == 128) return launch_patch_1x1<128>(p, s);
    if (p.Cin == 64) return launch_patch_1x1<64>(p, s);
    return launch_patch_1x1<32>(p, s);
  }
  if (small_map(p, dtype)) return dtype == PWR_BF16 ? launch_patch_small<bf16_t>(p, s) : launch_patch_small<float>(p, s);
  if (conv_wstat_applicable(p, dtype) || conv_wstat_narrow_applicable(p, dtype)) return launch_conv_wstat(p, nullptr, s);
  if (p.w_frag) return PWR_EINVAL;                   // a fragment-order pack and a launch that is not conv_wstat.hip's
  return dtype == PWR_BF16 ? launch_patch_t<bf16_t>(p, s) : launch_patch_t<float>(p, s);
}

}  // namespace pwr
