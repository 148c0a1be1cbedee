// Soft-argmax decoder, forward + backward, for gfx950 (wave64).
//
// Replaces the ~12 ATen ops of /root/reference/model.py:79-97 (PlaneRegression.forward after the
// conv head) and model.py:123-132 (DepthRegression.forward after the conv head) with one kernel
// per direction.  One workgroup owns one (b, j) map of N = P*P pixels:
//
//   forward   p = softmax_i(w_j * z_i)            | p = (relu(z)+1e-14)/sum        model.py:83-90
//             u = sum p*gx, v = sum p*gy          gx = (col-P/2)/(P-1), gy = (row-P/2)/(P-1)
//             d = sum (p*m)*(m*(D+L)) / (sum p*m + 1e-14)                          model.py:123-129
//   backward  closed form of what autograd derives for those lines (SURVEY.md 8 a-D).
//
// Memory plan (HBM bound): every [B,J,P,P] operand is read exactly once with 16-byte coalesced
// loads and kept in registers between the reduction and the element-wise tail; L and m
// ([B,1,P,P]) are shared by the J maps of a sample and are served by L2.  Algorithmic traffic:
// forward 12 B/pixel/map (z, D in; p out), backward 28 B (p, z, D, gH, gD in; gz, gD out).
#include <cstdlib>

#include "pwr_common.h"

namespace pwr {

#define PWR_DEC_EPS 1e-14f

__device__ __forceinline__ float grid_coord(int idx, int P) {
  // utils.py:28-34 rounded to fp32 (model.py:68): (idx - P//2)/(P-1), correctly rounded divide
  return __fdiv_rn((float)(idx - (P >> 1)), (float)(P - 1));
}

// Round 6 -- which map a workgroup owns.  L and m ([B,1,P,P]) are shared by the J maps of a sample and are meant to be served by L2, but
// workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share one: MI355X_MICROARCH.md, Workgroup dispatch) and every XCD has
// an L2 of its own: with map = blockIdx the J maps of a sample are spread over all eight, and each L2 fetches that sample's L and m from HBM
// again -- 8 x 16.8 MB at the C5 shape, the 1.12 x of round 5's forward counters.  With Bx > 0 the block index is read as (slot, xcd) and XCD
// x owns the samples b = 8 g + x: all J maps of a sample run on ONE XCD, back to back (grid = 8 ceil(B / 8) J; a block whose sample does
// not exist leaves).  Placement is a speed matter only: every map is computed by exactly one workgroup either way, same arithmetic.
__device__ __forceinline__ int dec_map(int bid, int J, int Bx) {
  if (Bx <= 0) return bid;
  const int xcd = bid & 7, slot = bid >> 3;
  const int g = slot / J, j = slot - g * J;
  const int b = g * 8 + xcd;
  return b < Bx ? b * J + j : -1;
}
// a 16-byte load of a tensor that is read ONCE (z, D, p, gH, gD): NTL = non-temporal (bypasses the vector L1, L2-served)
template <bool NTL>
__device__ __forceinline__ f32x4 dec_ld(const float* q) {
  if constexpr (NTL) return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(q));
  else return *reinterpret_cast<const f32x4*>(q);
}
template <bool NTL>
__device__ __forceinline__ void dec_st(float* q, const f32x4& v) {
  if constexpr (NTL) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(q));
  else *reinterpret_cast<f32x4*>(q) = v;
}

// ---------------------------------------------------------------------------------------------
// Register-resident path: N == NT*NV*4, P % 4 == 0 and (NT*4) % P == 0, so a thread's four
// columns are the same for all of its NV vectors.
// ---------------------------------------------------------------------------------------------
// MPW > 1 (round 6, the 128x128 maps): a workgroup owns MPW consecutive maps of one sample and software-pipelines them -- the logits of
// map i + 1 are requested before map i's block reductions and its p stores, so the CU (one workgroup of this size fits) has loads in flight
// during what used to be a load-free third of every map.  Same arithmetic per map.
// PFD: the next map's depth maps D are requested with its logits (another NV vectors of registers).
// LDS_EV (round 6 experiment): the exponentials wait for the block reductions in LDS (64 KiB per 128x128 map) instead of in registers, so
// that the kernel fits 128 registers and TWO workgroups share a CU (their load and store phases overlap by themselves).
template <int NT, int NV, int MINW = 1, bool NTL = false, bool PIPE = false, bool PFD = false, bool LDS_EV = false>
__global__ __launch_bounds__(NT, MINW) void decode_fwd_cached(const float* __restrict__ z, const float* __restrict__ D,
                                                        const float* __restrict__ L, const float* __restrict__ m,
                                                        const float* __restrict__ w, float* __restrict__ p_out,
                                                        float* __restrict__ uvd, int J, int P, int method, int Bx, int mpw_) {
  const int MPW = PIPE ? mpw_ : 1;            // maps per workgroup (consecutive maps of one sample)
  // ONE dependent memory phase: every thread exponentiates against its OWN maximum (all its logits are in registers), so the
  // loads of D, L and m are not held back behind a workgroup-wide max reduction; the two block reductions (max of the thread
  // maxima, then the sums rescaled by exp(m_thread - M)) sit back to back at the end with no memory access between them.
  //   p = exp(e - m_t) * exp(m_t - M) / S      (one extra rounding vs exp(e - M) / S: ~1e-7 relative)
  constexpr int NW = NT / 64;
  __shared__ float red[8 * NW];
  __shared__ __attribute__((aligned(16))) float evl[LDS_EV ? NT * NV * 4 : 4];
  const int map0 = dec_map(blockIdx.x, J / MPW, Bx) ;      // (group index: MPW maps per group)
  if (map0 < 0) return;
  const int N = P * P;
  const int tid = threadIdx.x;
  const int col0 = (tid * 4) % P;
  const int rows_per_step = (NT * 4) / P;
  const int row0 = (tid * 4) / P;

  f32x4 e[NV];
  f32x4 dcur[PFD ? NV : 1];
#pragma unroll
  for (int k = 0; k < NV; ++k) e[k] = dec_ld<NTL>(z + (size_t)map0 * MPW * N + (size_t)(k * NT + tid) * 4);
  if constexpr (PFD) {
#pragma unroll
    for (int k = 0; k < NV; ++k) dcur[k] = dec_ld<NTL>(D + (size_t)map0 * MPW * N + (size_t)(k * NT + tid) * 4);
  }
#pragma unroll 1
  for (int mi = 0; mi < MPW; ++mi) {
    const int map = map0 * MPW + mi, b = map / J, j = map - b * J;
    const size_t mo = (size_t)map * N, bo = (size_t)b * N;
    float mt = -INFINITY;
    const float wj = (method == 0) ? w[j] : 1.f;
    if (method == 0) {
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        e[k] *= wj;
        mt = fmaxf(mt, fmaxf(fmaxf(e[k].x, e[k].y), fmaxf(e[k].z, e[k].w)));
      }
    }

    float cs[4] = {0.f, 0.f, 0.f, 0.f};  // per-column sums of e
    float sv = 0.f, sm = 0.f, sd = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const size_t off = (size_t)(k * NT + tid) * 4;
      f32x4 dv;
      if constexpr (PFD) dv = dcur[k];
      else dv = dec_ld<NTL>(D + mo + off);
      f32x4 lv = *reinterpret_cast<const f32x4*>(L + bo + off);
      f32x4 mv = *reinterpret_cast<const f32x4*>(m + bo + off);
      f32x4 ev;
      if (method == 0) {
        ev.x = expf(e[k].x - mt); ev.y = expf(e[k].y - mt); ev.z = expf(e[k].z - mt); ev.w = expf(e[k].w - mt);
      } else {
        ev.x = fmaxf(e[k].x, 0.f) + PWR_DEC_EPS; ev.y = fmaxf(e[k].y, 0.f) + PWR_DEC_EPS;
        ev.z = fmaxf(e[k].z, 0.f) + PWR_DEC_EPS; ev.w = fmaxf(e[k].w, 0.f) + PWR_DEC_EPS;
      }
      if constexpr (LDS_EV) *reinterpret_cast<f32x4*>(evl + (size_t)(k * NT + tid) * 4) = ev;
      else e[k] = ev;
      cs[0] += ev.x; cs[1] += ev.y; cs[2] += ev.z; cs[3] += ev.w;
      const float gy = grid_coord(row0 + k * rows_per_step, P);
      sv += gy * ((ev.x + ev.y) + (ev.z + ev.w));
      f32x4 em = ev * mv;              // p*m (unnormalised)
      f32x4 mr = mv * (dv + lv);       // m*(D+L)
      sm += (em.x + em.y) + (em.z + em.w);
      sd += (em.x * mr.x + em.y * mr.y) + (em.z * mr.z + em.w * mr.w);
      // 128x128 maps (NV = 8): keep at most two iterations' loads in flight per thread, otherwise the scheduler hoists all 24 and
      // the kernel needs 144 VGPRs = one workgroup per CU; with <= 128 two workgroups overlap their phases
      if (MINW > 1 && (LDS_EV || (k & 1))) __builtin_amdgcn_sched_barrier(0);
    }
    // the next map's logits: requested here, consumed at the top of the next trip
    f32x4 en[PIPE ? NV : 1];
    if constexpr (PIPE) {
      if (mi + 1 < MPW) {
#pragma unroll
        for (int k = 0; k < NV; ++k) en[k] = dec_ld<NTL>(z + mo + (size_t)N + (size_t)(k * NT + tid) * 4);
        if constexpr (PFD) {       // (dcur is dead here: every vector of it was consumed in the loop above)
#pragma unroll
          for (int k = 0; k < NV; ++k) dcur[k] = dec_ld<NTL>(D + mo + (size_t)N + (size_t)(k * NT + tid) * 4);
        }
      }
    }
    float r[5];
    r[0] = (cs[0] + cs[1]) + (cs[2] + cs[3]);
    r[1] = (cs[0] * grid_coord(col0, P) + cs[1] * grid_coord(col0 + 1, P)) +
           (cs[2] * grid_coord(col0 + 2, P) + cs[3] * grid_coord(col0 + 3, P));
    r[2] = sv; r[3] = sm; r[4] = sd;
    float f = 1.f;
    if (method == 0) {
      const float M = block_max<NW>(mt, red);
      f = expf(mt - M);
#pragma unroll
      for (int i = 0; i < 5; ++i) r[i] *= f;
    }
    block_sum<5, NW>(r, red);
    const float inv = __fdiv_rn(1.f, r[0]);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      f32x4 pv, ek;
      if constexpr (LDS_EV) ek = *reinterpret_cast<const f32x4*>(evl + (size_t)(k * NT + tid) * 4);      // (the thread's own stores)
      else ek = e[k];
      pv.x = __fdiv_rn(ek.x * f, r[0]); pv.y = __fdiv_rn(ek.y * f, r[0]);
      pv.z = __fdiv_rn(ek.z * f, r[0]); pv.w = __fdiv_rn(ek.w * f, r[0]);
      dec_st<NTL>(p_out + mo + (size_t)(k * NT + tid) * 4, pv);
      if (LDS_EV) __builtin_amdgcn_sched_barrier(0);
    }
    if (tid == 0) {
      uvd[(size_t)map * 3 + 0] = r[1] * inv;
      uvd[(size_t)map * 3 + 1] = r[2] * inv;
      uvd[(size_t)map * 3 + 2] = __fdiv_rn(r[4] * inv, r[3] * inv + PWR_DEC_EPS);
    }
    if constexpr (PIPE) {
#pragma unroll
      for (int k = 0; k < NV; ++k) e[k] = en[k];
    }
  }
}

// Generic path: any P; three passes over the map (re-reads come from L2).
template <int NT>
__global__ __launch_bounds__(NT) void decode_fwd_generic(const float* __restrict__ z, const float* __restrict__ D,
                                                         const float* __restrict__ L, const float* __restrict__ m,
                                                         const float* __restrict__ w, float* __restrict__ p_out,
                                                         float* __restrict__ uvd, int J, int P, int method) {
  constexpr int NW = NT / 64;
  __shared__ float red[8 * NW];
  const int map = blockIdx.x, b = map / J, j = map - b * J;
  const int N = P * P;
  const size_t mo = (size_t)map * N, bo = (size_t)b * N;
  const int tid = threadIdx.x;
  const float wj = (method == 0) ? w[j] : 1.f;
  float mx = -INFINITY;
  if (method == 0) {
    for (int i = tid; i < N; i += NT) mx = fmaxf(mx, wj * z[mo + i]);
    mx = block_max<NW>(mx, red);
  }
  float r[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  for (int i = tid; i < N; i += NT) {
    const float zz = z[mo + i];
    const float ev = (method == 0) ? expf(wj * zz - mx) : fmaxf(zz, 0.f) + PWR_DEC_EPS;
    const int row = i / P, col = i - row * P;
    const float mv = m[bo + i];
    const float em = ev * mv;
    r[0] += ev;
    r[1] += ev * grid_coord(col, P);
    r[2] += ev * grid_coord(row, P);
    r[3] += em;
    r[4] += em * (mv * (D[mo + i] + L[bo + i]));
  }
  block_sum<5, NW>(r, red);
  for (int i = tid; i < N; i += NT) {
    const float zz = z[mo + i];
    const float ev = (method == 0) ? expf(wj * zz - mx) : fmaxf(zz, 0.f) + PWR_DEC_EPS;
    p_out[mo + i] = __fdiv_rn(ev, r[0]);
  }
  if (tid == 0) {
    const float inv = __fdiv_rn(1.f, r[0]);
    uvd[(size_t)map * 3 + 0] = r[1] * inv;
    uvd[(size_t)map * 3 + 1] = r[2] * inv;
    uvd[(size_t)map * 3 + 2] = __fdiv_rn(r[4] * inv, r[3] * inv + PWR_DEC_EPS);
  }
}

// ---------------------------------------------------------------------------------------------
// Backward.  With S = sum p*m + 1e-14 and d = uvd[..,2]:
//   g_p = gH + gu*gx + gv*gy + gd*m*(m*(D+L) - d)/S
//   softmax: g_z = w * p * (g_p - sum p*g_p);  g_w[j] += sum_i (g_z/w ... ) = A2 - A1*A3
//            with A1 = sum p*g_p, A2 = sum p*g_p*z, A3 = sum p*z
//   sum:     g_z = (g_p - A1)/T * [z>0],  T = sum (relu(z)+1e-14)
//   g_D = gD_in + gd * p * m*m / S
// gH / gD_in may be null (treated as zeros).  gw_part is [B*J] (reduced over b by decode_gw_reduce).
// ---------------------------------------------------------------------------------------------
// wave-uniform scalars (gU, uvd, w): with SC1 they are fetched by agent-scope relaxed atomic loads (global_load ... sc1: served by
// L2, bypassing the scalar data cache and the vector L1) instead of the s_load the compiler picks for uniform read-only addresses
template <bool SC1>
__device__ __forceinline__ float uniform_load(const float* p) {
  if constexpr (SC1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *p;
}

// Single pass: with A = gH + gu*gx + gv*gy and Bm = m*(m*(D+L) - d) the incoming gradient is g_p = A + (gd/S)*Bm, so every sum
// the backward needs splits into sums that do NOT depend on S:
//   S = sum p*m, T1 = sum p*A, T2 = sum p*Bm, T3 = sum p*A*z, T4 = sum p*Bm*z, T5 = sum p*z   ->   ONE block reduction
//   A1 = T1 + gdS*T2,  g_w = (T3 + gdS*T4) - A1*T5,  g_z = w*p*(A + gdS*Bm - A1),  g_D = gD + gdS*p*m^2
// and all operands of a map are loaded in one dependent phase (the two-reduction form read m, D, L, z behind the first barrier).
// RELOAD (softmax only; used for the 128x128 maps where NV = 8): only p and Bm stay in registers, gH and m are fetched again
// (L2) in the tail -- 100 instead of 256 VGPRs, so that two 512-thread workgroups share a CU and overlap their phases.
template <int NT, int NV, bool RELOAD = false, bool SC1 = false, bool NTL = false>
__global__ __launch_bounds__(NT, RELOAD ? (2 * NT) / 256 : 1) void decode_bwd_cached(const float* __restrict__ p, const float* __restrict__ z,
                                                        const float* __restrict__ D, const float* __restrict__ L,
                                                        const float* __restrict__ m, const float* __restrict__ w,
                                                        const float* __restrict__ uvd, const float* __restrict__ gH,
                                                        const float* __restrict__ gDin, const float* __restrict__ gU,
                                                        float* __restrict__ gz, float* __restrict__ gDout,
                                                        float* __restrict__ gw_part, int J, int P, int method, int Bx) {
  constexpr int NW = NT / 64;
  constexpr int NK = RELOAD ? 1 : NV;      // vectors of A / p*m^2 kept in registers
  __shared__ float red[8 * NW];
  const int map = dec_map(blockIdx.x, J, Bx);
  if (map < 0) return;
  const int b = map / J, j = map - b * J;
  const int N = P * P;
  const size_t mo = (size_t)map * N, bo = (size_t)b * N;
  const int tid = threadIdx.x;
  const int col0 = (tid * 4) % P;
  const int rows_per_step = (NT * 4) / P;
  const int row0 = (tid * 4) / P;
  const float gu = uniform_load<SC1>(gU + (size_t)map * 3 + 0), gv = uniform_load<SC1>(gU + (size_t)map * 3 + 1),
              gd = uniform_load<SC1>(gU + (size_t)map * 3 + 2);
  const float d = uniform_load<SC1>(uvd + (size_t)map * 3 + 2);
  const float wj = (method == 0) ? uniform_load<SC1>(w + j) : 1.f;
  const float gxc[4] = {gu * grid_coord(col0, P), gu * grid_coord(col0 + 1, P), gu * grid_coord(col0 + 2, P),
                        gu * grid_coord(col0 + 3, P)};

  f32x4 pv[NV], Bv[NV], Av[NK], pm2[NK];
  float r[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // S (without eps), T1..T5, T (sum method)
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const size_t off = (size_t)(k * NT + tid) * 4;
    const f32x4 pk = dec_ld<NTL>(p + mo + off);
    const f32x4 mv = *reinterpret_cast<const f32x4*>(m + bo + off);
    const f32x4 dv = dec_ld<NTL>(D + mo + off);
    const f32x4 lv = *reinterpret_cast<const f32x4*>(L + bo + off);
    const f32x4 zv = dec_ld<NTL>(z + mo + off);
    const float gyv = gv * grid_coord(row0 + k * rows_per_step, P);
    f32x4 A;
    A.x = gxc[0] + gyv; A.y = gxc[1] + gyv; A.z = gxc[2] + gyv; A.w = gxc[3] + gyv;
    if (gH) A += dec_ld<NTL>(gH + mo + off);
    f32x4 Bm;
    Bm.x = mv.x * (mv.x * (dv.x + lv.x) - d); Bm.y = mv.y * (mv.y * (dv.y + lv.y) - d);
    Bm.z = mv.z * (mv.z * (dv.z + lv.z) - d); Bm.w = mv.w * (mv.w * (dv.w + lv.w) - d);
    const f32x4 pmv = pk * mv, pA = pk * A, pB = pk * Bm;
    r[0] += (pmv.x + pmv.y) + (pmv.z + pmv.w);
    r[1] += (pA.x + pA.y) + (pA.z + pA.w);
    r[2] += (pB.x + pB.y) + (pB.z + pB.w);
    r[3] += (pA.x * zv.x + pA.y * zv.y) + (pA.z * zv.z + pA.w * zv.w);
    r[4] += (pB.x * zv.x + pB.y * zv.y) + (pB.z * zv.z + pB.w * zv.w);
    r[5] += (pk.x * zv.x + pk.y * zv.y) + (pk.z * zv.z + pk.w * zv.w);
    pv[k] = pk; Bv[k] = Bm;
    if (RELOAD) __builtin_amdgcn_sched_barrier(0);     // one iteration's six loads in flight (see the forward kernel)
    if constexpr (!RELOAD) {
      Av[k] = A; pm2[k] = pmv * mv;
      if (method != 0) {
        // sum-normalisation: g_z = (g_p - A1)/T * [z>0]; p itself is not needed in the tail (p*m^2 is kept), so pv carries the mask
        r[6] += (fmaxf(zv.x, 0.f) + PWR_DEC_EPS) + (fmaxf(zv.y, 0.f) + PWR_DEC_EPS) + (fmaxf(zv.z, 0.f) + PWR_DEC_EPS) +
                (fmaxf(zv.w, 0.f) + PWR_DEC_EPS);
        f32x4 msk;
        msk.x = zv.x > 0.f ? 1.f : 0.f; msk.y = zv.y > 0.f ? 1.f : 0.f;
        msk.z = zv.z > 0.f ? 1.f : 0.f; msk.w = zv.w > 0.f ? 1.f : 0.f;
        pv[k] = msk;
      }
    }
  }
  block_sum<7, NW>(r, red);
  const float S = r[0] + PWR_DEC_EPS;
  const float gdS = __fdiv_rn(gd, S);
  const float A1 = r[1] + gdS * r[2];
  const float invT = (method != 0) ? __fdiv_rn(1.f, r[6]) : 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const size_t off = (size_t)(k * NT + tid) * 4;
    f32x4 A, q2;
    if constexpr (RELOAD) {
      const float gyv = gv * grid_coord(row0 + k * rows_per_step, P);
      A.x = gxc[0] + gyv; A.y = gxc[1] + gyv; A.z = gxc[2] + gyv; A.w = gxc[3] + gyv;
      if (gH) A += *reinterpret_cast<const f32x4*>(gH + mo + off);
      const f32x4 mv = *reinterpret_cast<const f32x4*>(m + bo + off);
      q2 = (pv[k] * mv) * mv;
    } else {
      A = Av[k]; q2 = pm2[k];
    }
    const f32x4 g = A + gdS * Bv[k];
    if (RELOAD && (k & 1)) __builtin_amdgcn_sched_barrier(0);
    f32x4 gzv;
    if (method == 0) gzv = wj * (pv[k] * (g - A1));
    else gzv = pv[k] * ((g - A1) * invT);
    dec_st<NTL>(gz + mo + off, gzv);
    f32x4 gdv = gdS * q2;
    if (gDin) gdv += dec_ld<NTL>(gDin + mo + off);
    dec_st<NTL>(gDout + mo + off, gdv);
  }
  if (tid == 0 && gw_part) gw_part[map] = (method == 0) ? ((r[3] + gdS * r[4]) - A1 * r[5]) : 0.f;
}

template <int NT>
__global__ __launch_bounds__(NT) void decode_bwd_generic(const float* __restrict__ p, const float* __restrict__ z,
                                                         const float* __restrict__ D, const float* __restrict__ L,
                                                         const float* __restrict__ m, const float* __restrict__ w,
                                                         const float* __restrict__ uvd, const float* __restrict__ gH,
                                                         const float* __restrict__ gDin, const float* __restrict__ gU,
                                                         float* __restrict__ gz, float* __restrict__ gDout,
                                                         float* __restrict__ gw_part, int J, int P, int method) {
  constexpr int NW = NT / 64;
  __shared__ float red[8 * NW];
  const int map = blockIdx.x, b = map / J, j = map - b * J;
  const int N = P * P;
  const size_t mo = (size_t)map * N, bo = (size_t)b * N;
  const int tid = threadIdx.x;
  const float gu = gU[(size_t)map * 3 + 0], gv = gU[(size_t)map * 3 + 1], gd = gU[(size_t)map * 3 + 2];
  const float d = uvd[(size_t)map * 3 + 2];
  const float wj = (method == 0) ? w[j] : 1.f;
  float r1[2] = {0.f, 0.f};
  for (int i = tid; i < N; i += NT) {
    r1[0] += p[mo + i] * m[bo + i];
    if (method != 0) r1[1] += fmaxf(z[mo + i], 0.f) + PWR_DEC_EPS;
  }
  block_sum<2, NW>(r1, red);
  const float S = r1[0] + PWR_DEC_EPS;
  const float gdS = __fdiv_rn(gd, S);
  float r2[3] = {0.f, 0.f, 0.f};
  for (int i = tid; i < N; i += NT) {
    const int row = i / P, col = i - row * P;
    const float mv = m[bo + i], pp = p[mo + i], zz = z[mo + i];
    float g = gu * grid_coord(col, P) + gv * grid_coord(row, P) + gdS * mv * (mv * (D[mo + i] + L[bo + i]) - d);
    if (gH) g += gH[mo + i];
    r2[0] += pp * g; r2[1] += pp * g * zz; r2[2] += pp * zz;
  }
  block_sum<3, NW>(r2, red);
  const float A1 = r2[0];
  const float invT = (method != 0) ? __fdiv_rn(1.f, r1[1]) : 0.f;
  for (int i = tid; i < N; i += NT) {
    const int row = i / P, col = i - row * P;
    const float mv = m[bo + i], pp = p[mo + i], zz = z[mo + i];
    float g = gu * grid_coord(col, P) + gv * grid_coord(row, P) + gdS * mv * (mv * (D[mo + i] + L[bo + i]) - d);
    if (gH) g += gH[mo + i];
    gz[mo + i] = (method == 0) ? wj * (pp * (g - A1)) : (zz > 0.f ? (g - A1) * invT : 0.f);
    float gdv = gdS * pp * mv * mv;
    if (gDin) gdv += gDin[mo + i];
    gDout[mo + i] = gdv;
  }
  if (tid == 0 && gw_part) gw_part[map] = (method == 0) ? (r2[1] - A1 * r2[2]) : 0.f;
}

// gw[j] (+)= sum_b gw_part[b*J + j]   (deterministic order)
__global__ void decode_gw_reduce(const float* __restrict__ gw_part, float* __restrict__ gw, int B, int J,
                                 int accumulate) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= J) return;
  float s = 0.f;
  for (int b = 0; b < B; ++b) s += gw_part[(size_t)b * J + j];
  gw[j] = accumulate ? gw[j] + s : s;
}

}  // namespace pwr

// ---------------------------------------------------------------------------------------------
// C ABI (declared in include/pwr.h)
// ---------------------------------------------------------------------------------------------
// XCD-aware map order (dec_map) and non-temporal accesses of the once-read tensors: switches of the debug build (tools/bench_decoder.py
// A/B); the product's values are the measured best (decoder)
static inline int dec_grid(int B, int J, int* Bx, int mpw = 1) {
  static const int xcd = PWR_DBG_ENV("PWR_DEC_XCD", 1);
  *Bx = xcd ? B : 0;
  return xcd ? 8 * ((B + 7) / 8) * (J / mpw) : B * (J / mpw);
}
// Measured (MI355X, tools/bench_decoder.py, three interleaved runs each; profiles/r6_experiments.md): the XCD-aware order takes the forward's
// fetched bytes at the C5 shape from 853 to 737 MB (721 algorithmic) and the 64x64 backward from 27.9 to 24.9 us (C3 shape); non-temporal
// accesses of the once-read tensors: forward -4 ... -7 % at every shape, backward -8 % on the 128x128 maps and +6 % on the 64x64 ones
// (their operands were written by the launches just before and are still cached).  PWR_DEC_NT bits: 1 forward, 2 backward 128x128,
// 4 backward 64x64.
extern "C" int pwr_decode_fwd(const float* z, const float* D, const float* L, const float* m, const float* w,
                              float* p_out, float* uvd_out, int B, int J, int P, int method, void* stream) {
  if (B <= 0 || J <= 0 || P <= 1 || (method == 0 && !w)) return -1;
  hipStream_t s = (hipStream_t)stream;
  const int N = P * P, maps = B * J;
  int Bx = 0;
  const int grid = dec_grid(B, J, &Bx);
  static const int nt = PWR_DBG_ENV("PWR_DEC_NT", 3) & 1;
  if (P % 4 == 0 && N == 256 * 4 * 4 && 1024 % P == 0) {
    if (nt) hipLaunchKernelGGL((pwr::decode_fwd_cached<256, 4, 1, true>), dim3(grid), dim3(256), 0, s, z, D, L, m, w, p_out, uvd_out, J, P, method, Bx, 1);
    else hipLaunchKernelGGL((pwr::decode_fwd_cached<256, 4>), dim3(grid), dim3(256), 0, s, z, D, L, m, w, p_out, uvd_out, J, P, method, Bx, 1);
  } else if (P % 4 == 0 && N == 256 * 4 && 1024 % P == 0)
    hipLaunchKernelGGL((pwr::decode_fwd_cached<256, 1>), dim3(grid), dim3(256), 0, s, z, D, L, m, w, p_out, uvd_out, J, P, method, Bx, 1);
  else if (P % 4 == 0 && N == 512 * 8 * 4 && 2048 % P == 0) {
    // 128x128 maps.  A/B switch (tools/bench_decoder.py), measured at B=128, J=42 on MI355X (profiles/r2_decoder_bench.jsonl):
    // 0 = 512 threads x 8 vectors, 144 VGPRs, one workgroup per CU: 4.28 TB/s (default); 1 = 1024 threads x 4 vectors: 4.24 TB/s;
    // 2 = 512 x 8 held to 128 VGPRs (two workgroups per CU, a few spilled registers): 3.87 TB/s
    static const int v = PWR_DBG_ENV("PWR_DEC_FWD128", 0);
    // round 6: MPW maps per workgroup, software-pipelined (PWR_DEC_PIPE, 0 = off) -- where there are enough maps to keep every CU busy
    // for several rounds anyway
    // (pipe = 2: the next map's depth maps prefetched too, 223 registers: no faster than 1 -- 233.4 against 235.5 us, within the noise)
    static const int pipe = PWR_DBG_ENV("PWR_DEC_PIPE", 1);
    // maps per workgroup: a divisor of J; cost model = rounds of workgroups over the 256 CUs x maps per workgroup (PWR_DEC_MPW, debug build:
    // force a value)
    int mpw = 1;
    if (pipe && v == 0 && maps >= 8 * 256) {
      // (the SMALLEST such divisor > 1: the groups of a sample run side by side on one XCD and share its L and m in L2; with 21 maps per
      // workgroup -- two groups per sample -- L and m were evicted between maps and the forward fetched 996 instead of 737 MB, for the same time)
      const int Bp = 8 * ((B + 7) / 8);
      auto cost = [&](int q) { return (((long long)Bp * (J / q) + 255) / 256) * q; };
      long long best = -1;
      int bq = 1;
      for (int q = 2; q <= J; ++q) {
        if (J % q) continue;
        if (best < 0 || cost(q) < best) { best = cost(q); bq = q; }
      }
      if (best >= 0 && best <= cost(1)) mpw = bq;
      static const int force = PWR_DBG_ENV("PWR_DEC_MPW", 0);
      if (force > 0 && J % force == 0) mpw = force;
    }
    const int gp = dec_grid(B, J, &Bx, mpw);
    if (v == 1 && 4096 % P == 0)
      hipLaunchKernelGGL((pwr::decode_fwd_cached<1024, 4>), dim3(grid), dim3(1024), 0, s, z, D, L, m, w, p_out, uvd_out, J, P, method, Bx, 1);
    else if (v == 2)
      hipLaunchKernelGGL((pwr::decode_fwd_cached<512, 8, 4>), dim3(grid), dim3(512), 0, s, z, D, L, m, w, p_out, uvd_out, J, P, method, Bx, 1);
#ifdef PWR_DEBUG_BUILD
    // (round 6 experiments, debug build only -- both SLOWER than the pipelined one-workgroup-per-CU form, 229 us: 4 = exponentials staged in
    // LDS, 128 registers, two 512-thread workgroups per CU: 262 us; 3 = 256 threads x 16 vectors, two workgroups per CU: 253 us)
    else if (v == 4)
      hipLaunchKernelGGL((pwr::decode_fwd_cached<512, 8, 4, true, false, false, true>), dim3(grid), dim3(512), 0, s, z, D, L, m, w, p_out, uvd_out, J, P, method, Bx, 1);
    else if (v == 3 && 1024 % P == 0)
      hipLaunchKernelGGL((pwr::decode_fwd_cached<256, 16, 2, true>), dim3(grid), dim3(256), 0, s, z, D, L, m, w, p_out, uvd_out, J, P, method, Bx, 1);
#endif
    else if (mpw > 1 && nt && pipe == 2)
      hipLaunchKernelGGL((pwr::decode_fwd_cached<512, 8, 1, true, true, true>), dim3(gp), dim3(512), 0, s, z, D, L, m, w, p_out, uvd_out, J, P, method, Bx, mpw);
    else if (mpw > 1 && nt)
      hipLaunchKernelGGL((pwr::decode_fwd_cached<512, 8, 1, true, true>), dim3(gp), dim3(512), 0, s, z, D, L, m, w, p_out, uvd_out, J, P, method, Bx, mpw);
    else if (mpw > 1)
      hipLaunchKernelGGL((pwr::decode_fwd_cached<512, 8, 1, false, true>), dim3(gp), dim3(512), 0, s, z, D, L, m, w, p_out, uvd_out, J, P, method, Bx, mpw);
    else if (nt)
      hipLaunchKernelGGL((pwr::decode_fwd_cached<512, 8, 1, true>), dim3(grid), dim3(512), 0, s, z, D, L, m, w, p_out, uvd_out, J, P, method, Bx, 1);
    else
      hipLaunchKernelGGL((pwr::decode_fwd_cached<512, 8>), dim3(grid), dim3(512), 0, s, z, D, L, m, w, p_out, uvd_out, J, P, method, Bx, 1);
  }
  else
    hipLaunchKernelGGL((pwr::decode_fwd_generic<256>), dim3(maps), dim3(256), 0, s, z, D, L, m, w, p_out, uvd_out, J, P, method);
  return (int)hipGetLastError();
}

extern "C" int pwr_decode_bwd(const float* p, const float* z, const float* D, const float* L, const float* m,
                              const float* w, const float* uvd, const float* gH, const float* gD_in,
                              const float* gU, float* gz_out, float* gD_out, float* gw_part, int B, int J, int P,
                              int method, void* stream) {
  if (B <= 0 || J <= 0 || P <= 1 || (method == 0 && !w)) return -1;
  hipStream_t s = (hipStream_t)stream;
  const int N = P * P, maps = B * J;
  int Bx = 0;
  const int grid = dec_grid(B, J, &Bx);
  static const int ntf = PWR_DBG_ENV("PWR_DEC_NT", 3);
  const int nt = P >= 128 ? (ntf & 2) : (ntf & 4);      // (see pwr_decode_fwd)
  static const int sc1 = PWR_DBG_ENV("PWR_DEC_SCALAR_SC1", 0);   // experiment switch (tools/race_hunt.py)
  if (sc1 && P % 4 == 0 && N == 256 * 4 * 4 && 1024 % P == 0)
    hipLaunchKernelGGL((pwr::decode_bwd_cached<256, 4, false, true>), dim3(grid), dim3(256), 0, s, p, z, D, L, m, w, uvd, gH, gD_in, gU, gz_out, gD_out, gw_part, J, P, method, Bx);
  else if (P % 4 == 0 && N == 256 * 4 * 4 && 1024 % P == 0) {
    if (nt) hipLaunchKernelGGL((pwr::decode_bwd_cached<256, 4, false, false, true>), dim3(grid), dim3(256), 0, s, p, z, D, L, m, w, uvd, gH, gD_in, gU, gz_out, gD_out, gw_part, J, P, method, Bx);
    else hipLaunchKernelGGL((pwr::decode_bwd_cached<256, 4>), dim3(grid), dim3(256), 0, s, p, z, D, L, m, w, uvd, gH, gD_in, gU, gz_out, gD_out, gw_part, J, P, method, Bx);
  } else if (P % 4 == 0 && N == 256 * 4 && 1024 % P == 0)
    hipLaunchKernelGGL((pwr::decode_bwd_cached<256, 1>), dim3(grid), dim3(256), 0, s, p, z, D, L, m, w, uvd, gH, gD_in, gU, gz_out, gD_out, gw_part, J, P, method, Bx);
  else if (P % 4 == 0 && N == 512 * 8 * 4 && 2048 % P == 0) {
    // 128x128 maps.  A/B switch, measured like the forward: 0 = 512 x 8 with everything in registers (204 VGPRs, one workgroup per
    // CU): 4.74 TB/s (default; the two-reduction form of round 1: 3.04); 1 = 1024 x 4 (108 VGPRs, one 16-wave workgroup per CU): 4.38;
    // 2 = 512 x 8 RELOAD (softmax only; 128 VGPRs with spills, two workgroups per CU): 2.51
    static const int v = PWR_DBG_ENV("PWR_DEC_BWD128", 0);
    if (v == 1 && 4096 % P == 0)
      hipLaunchKernelGGL((pwr::decode_bwd_cached<1024, 4>), dim3(grid), dim3(1024), 0, s, p, z, D, L, m, w, uvd, gH, gD_in, gU, gz_out, gD_out, gw_part, J, P, method, Bx);
    else if (v == 2 && method == 0)
      hipLaunchKernelGGL((pwr::decode_bwd_cached<512, 8, true>), dim3(grid), dim3(512), 0, s, p, z, D, L, m, w, uvd, gH, gD_in, gU, gz_out, gD_out, gw_part, J, P, method, Bx);
    else if (nt)
      hipLaunchKernelGGL((pwr::decode_bwd_cached<512, 8, false, false, true>), dim3(grid), dim3(512), 0, s, p, z, D, L, m, w, uvd, gH, gD_in, gU, gz_out, gD_out, gw_part, J, P, method, Bx);
    else
      hipLaunchKernelGGL((pwr::decode_bwd_cached<512, 8>), dim3(grid), dim3(512), 0, s, p, z, D, L, m, w, uvd, gH, gD_in, gU, gz_out, gD_out, gw_part, J, P, method, Bx);
  }
  else
    hipLaunchKernelGGL((pwr::decode_bwd_generic<256>), dim3(maps), dim3(256), 0, s, p, z, D, L, m, w, uvd, gH, gD_in, gU, gz_out, gD_out, gw_part, J, P, method);
  return (int)hipGetLastError();
}

extern "C" int pwr_decode_gw_reduce(const float* gw_part, float* gw, int B, int J, int accumulate, void* stream) {
  hipLaunchKernelGGL(pwr::decode_gw_reduce, dim3((J + 63) / 64), dim3(64), 0, (hipStream_t)stream, gw_part, gw, B, J,
                     accumulate);
  return (int)hipGetLastError();
}
