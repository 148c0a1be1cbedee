// Shared by the implicit-GEMM convolution kernels (conv_mfma.hip: universal; conv_patch.hip: LDS-resident input patch).
#pragma once
#include <type_traits>
#include "pwr_common.h"

namespace pwr {

struct ConvParams {
  const void* x;          // [B,H,W,Cin] T   (forward: input; dgrad: dy)
  const void* w;          // packed weights, see pack_weights kernel
  const float* bias;      // [Cout] or null
  const float* in_norm;   // [4][B][Cin] = mean, rstd, scale, beta of the input's norm (or null): NR prologue
  const void* residual;   // [B,Ho,Wo,Cout] T or null (added in the epilogue)
  void* y;                // [B,Ho,Wo,Cout] T or null
  float* y_nchw;          // [B,Cout,Ho,Wo] fp32 or null
  int B, H, W, Cin, Ho, Wo, Cout, CoutPad;
  int ksize, stride, pad, mode, relu_in, KCH, M;
  long long* stamps = nullptr;   // debugging aid (pwr_debug_set_stamps): 8 x int64 per workgroup = phase time stamps + HW ids
  int dbg_delay = 0;             // debug build (pwr_debug_set_delay): workgroups in an odd wave slot sleep ~dbg_delay cycles before staging
  // ---- optional per-channel column statistics of the OUTPUT tile, written by the epilogue (one slab entry per workgroup;
  // a workgroup's 128 output pixels lie in one sample).  nb_partial: [(b*chunks + chunk)*2 + {0,1}][Cout] fp32.
  // st_partial: [(b*chunks + chunk)*3 + {0,1,2}][Cout] = sum (v - k), sum (v - k)^2, k of the stored output v, with the shift
  //             k[c] = the tile's first output row (so the sums never cancel): the statistics of the norm that FOLLOWS this
  //             conv (replaces norm_partial_kernel; combined over chunks by norm_finalize_chunks_kernel).
  // nb_partial: this launch is a data gradient producing g = dL/d relu(norm(y)); sums of gm and gm * xhat for the norm
  //             backward of y (nb_y, nb_state = [4][B][Cout]): replaces norm_bwd_partial_kernel.
  float* st_partial = nullptr;
  const void* nb_y = nullptr;
  const float* nb_state = nullptr;
  float* nb_partial = nullptr;
  int nb_relu = 1;
  // slab row of a workgroup's statistics = b * st_nchunks + st_chunk0 + (its tile index inside the image); st_nchunks == 0: the tiles per
  // image of this launch.  The four parity-class launches of a stride-2 data gradient (conv_patch.hip, GEO 2 - 5) write one slab between
  // them: 4 x tiles rows per sample, class c at rows c * tiles ...
  int st_nchunks = 0, st_chunk0 = 0;
  // FOLD (round 6, conv_patch.hip's FB forms): x is the RAW gradient g = dL/d relu(norm(fb_y)).  The staging computes the norm backward's
  // dy from g, fb_y, fb_state ([4][B][Cin]) and the sums of fb_partial (fb_pchunks slab rows of [2][Cin] per sample) on the way into LDS --
  // the values pwr_norm_bwd_apply_from_partial would have written, bit for bit -- and stores the tile's own pixels of it to fb_dy (the
  // weight gradient's operand): the apply launch between two data gradients is gone.
  const void* fb_y = nullptr;
  const float* fb_state = nullptr;
  const float* fb_partial = nullptr;
  void* fb_dy = nullptr;
  int fb_pchunks = 0, fb_relu = 1;
  int epi16 = 1;            // (debug build: 0 = the two-pass fp32 epilogue also for the 16x16x32 tile)
  int w_frag = 0;           // the pack is in conv_wstat.hip's fragment order (PackDesc::order 1; the caller passed the pack address with bit 0 set)
};

struct WgradParams {
  const void* x;          // [B,H,W,Cin] T  forward input (pre-NR)
  const void* dy;         // [B,Ho,Wo,Cout] T
  const float* in_norm;   // NR prologue of the forward conv (or null), [4][B][Cin]
  float* slab;            // [S][taps][CinPad128][CoutPad] fp32 partials
  int B, H, W, Cin, Ho, Wo, Cout, CoutPad, CinPad;
  int ksize, stride, pad, relu_in, M, S, steps_per_split;
  int dbg = 0;            // debug build only: elimination bits of conv_wgrad3d_kernel (1 no MFMA, 2 no fragment reads, 4 no DMA after the prologue, 8 no slab stores)
  long long* stamps = nullptr;   // debug build only (conv_wgrad_ws.hip): per wave {cycles in barriers, cycles in DMA waits, cycles in the loop, steps}
};

template <typename T> struct Mma;
template <> struct Mma<bf16_t> { static constexpr int KE = 32; static constexpr int EP = 8; };
template <> struct Mma<float> { static constexpr int KE = 16; static constexpr int EP = 4; };

__device__ __forceinline__ int lds_off(int row, int slot) { return row * 64 + (((slot ^ (row >> 2)) & 3) << 4); }

template <typename T, int MR, int NR>
__device__ __forceinline__ void mma_tile(const char* lA, const char* lB, int a_row0, int b_row0, int lane,
                                         f32x16 (&acc)[MR][NR]) {
  typedef typename Vec16<T>::type V;
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int ss = 0; ss < 2; ++ss) {
    V a[MR], b[NR];
#pragma unroll
    for (int i = 0; i < MR; ++i) a[i] = *reinterpret_cast<const V*>(lA + lds_off(a_row0 + i * 32 + r, 2 * ss + h));
#pragma unroll
    for (int j = 0; j < NR; ++j) b[j] = *reinterpret_cast<const V*>(lB + lds_off(b_row0 + j * 32 + r, 2 * ss + h));
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        if constexpr (sizeof(T) == 2) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
        }
      }
  }
}

// v = relu?((v - mean)*scale + beta) on one 16-byte vector; st points at mean[b][c0] of a [4][B][C] norm state
// (mean, rstd, scale = gamma*rstd, beta).  Subtracting the mean first keeps the cancellation exact-ish, like
// ATen's (x - mean) * invstd * gamma + beta.
template <typename T>
__device__ __forceinline__ typename Vec16<T>::type nr_transform(typename Vec16<T>::type v, const float* st, size_t plane,
                                                                int relu) {
  constexpr int EP = Mma<T>::EP;
  typename Vec16<T>::type o;
#pragma unroll
  for (int e = 0; e < EP; ++e) {
    float f = fmaf(Elem<T>::to_f(v[e]) - st[e], st[2 * plane + e], st[3 * plane + e]);
    if (relu) f = fmaxf(f, 0.f);
    o[e] = Elem<T>::from_f(f);
  }
  return o;
}

// Column statistics of a conv epilogue (see ConvParams::st_partial / nb_partial).  Every thread of the epilogue's copy loop
// owns one 16-byte channel slot (EP channels) for all of its rows: add() per stored vector, finish() once per workgroup.
template <typename T>
struct EpiStats {
  static constexpr int EP = Mma<T>::EP;
  typedef typename Vec16<T>::type V;
  float s1[EP], s2[EP], a0[EP], a1[EP], a2[EP], a3[EP];
  int kind = 0;   // 0 off, 1 forward statistics, 2 norm-backward sums

  // b: sample of this workgroup's tile; n: first channel of this thread's slot (inactive if n >= Cout)
  __device__ __forceinline__ void init(const ConvParams& p, int b, int n) {
    kind = p.st_partial ? 1 : (p.nb_partial ? 2 : 0);
#pragma unroll
    for (int e = 0; e < EP; ++e) { s1[e] = 0.f; s2[e] = 0.f; a0[e] = 0.f; a1[e] = 0.f; a2[e] = 0.f; a3[e] = 0.f; }
    if (n >= p.Cout) return;
    if (kind == 2) {
      const size_t plane = (size_t)p.B * p.Cout, c = (size_t)b * p.Cout + n;
#pragma unroll
      for (int e = 0; e < EP; ++e) { a0[e] = p.nb_state[c + e]; a1[e] = p.nb_state[plane + c + e]; a2[e] = p.nb_state[2 * plane + c + e]; a3[e] = p.nb_state[3 * plane + c + e]; }
    }
  }
  // forward statistics: shift = (approximately) the tile's first output row; e0 = its fp32 accumulators for this slot
  __device__ __forceinline__ void set_shift(const ConvParams& p, const float* e0, int n) {
    if (kind != 1 || n >= p.Cout) return;
#pragma unroll
    for (int e = 0; e < EP; ++e) a0[e] = e0[e] + (p.bias ? p.bias[n + e] : 0.f);
  }
  // o: the vector just stored at output row m, channels n..n+EP-1
  __device__ __forceinline__ void add(const ConvParams& p, const V& o, size_t m, int n) {
    if (kind == 1) {
#pragma unroll
      for (int e = 0; e < EP; ++e) { const float d = Elem<T>::to_f(o[e]) - a0[e]; s1[e] += d; s2[e] = fmaf(d, d, s2[e]); }
    } else if (kind == 2) {
      const V yv = *reinterpret_cast<const V*>(reinterpret_cast<const T*>(p.nb_y) + m * p.Cout + n);
#pragma unroll
      for (int e = 0; e < EP; ++e) {
        const float yy = Elem<T>::to_f(yv[e]);
        float gg = Elem<T>::to_f(o[e]);
        if (p.nb_relu && !(fmaf(yy - a0[e], a2[e], a3[e]) > 0.f)) gg = 0.f;
        s1[e] += gg;
        s2[e] = fmaf(gg, (yy - a0[e]) * a1[e], s2[e]);
      }
    }
  }
  // the same with the forward activations of the vector (kind 2) already in registers
  __device__ __forceinline__ void add_pre(const ConvParams& p, const V& o, const V& yv) {
    if (kind == 1) {
#pragma unroll
      for (int e = 0; e < EP; ++e) { const float d = Elem<T>::to_f(o[e]) - a0[e]; s1[e] += d; s2[e] = fmaf(d, d, s2[e]); }
    } else if (kind == 2) {
#pragma unroll
      for (int e = 0; e < EP; ++e) {
        const float yy = Elem<T>::to_f(yv[e]);
        float gg = Elem<T>::to_f(o[e]);
        if (p.nb_relu && !(fmaf(yy - a0[e], a2[e], a3[e]) > 0.f)) gg = 0.f;
        s1[e] += gg;
        s2[e] = fmaf(gg, (yy - a0[e]) * a1[e], s2[e]);
      }
    }
  }
  // CPR = 16-byte slots per tile row (threads tid % CPR == slot share a slot), NT threads.  `lds`: >= (NT/64)*CPR*2*EP floats,
  // free for use (call after the epilogue's last barrier).  n0: first channel of the tile.
  template <int CPR, int NT>
  __device__ __forceinline__ void finish(const ConvParams& p, float* lds, int b, int chunk, int nchunks, int n0) {
    if (kind == 0) return;   // (uniform)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, slot = tid % CPR;
    // (lane ^ o through DPP / v_permlane swaps, pwr_common.h: the same sums as the __shfl_xor loop -- o = CPR, 2 CPR, ... 32 -- without its
    // ds_bpermute_b32 round trips at the tail of every workgroup)
    auto step = [&](auto O) __attribute__((always_inline)) {
      constexpr int o = decltype(O)::value;
      if constexpr (o >= CPR) {
#pragma unroll
        for (int e = 0; e < EP; ++e) { s1[e] = lane_xor_add<o>(s1[e]); s2[e] = lane_xor_add<o>(s2[e]); }
      }
    };
    static_assert(CPR == 2 || CPR == 4 || CPR == 8 || CPR == 16 || CPR == 32, "slots per tile row");
    step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 8>{});
    step(std::integral_constant<int, 16>{}); step(std::integral_constant<int, 32>{});
    if (lane < CPR) {
#pragma unroll
      for (int e = 0; e < EP; ++e) { lds[((wid * 2 + 0) * CPR + slot) * EP + e] = s1[e]; lds[((wid * 2 + 1) * CPR + slot) * EP + e] = s2[e]; }
    }
    __syncthreads();
    float* out = kind == 1 ? p.st_partial + ((size_t)(b * nchunks + chunk) * 3) * p.Cout : p.nb_partial + ((size_t)(b * nchunks + chunk) * 2) * p.Cout;
    for (int i = tid; i < 2 * CPR * EP; i += NT) {
      const int which = i / (CPR * EP), c = i - which * (CPR * EP);
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < NT / 64; ++w) t += lds[((w * 2 + which) * CPR) * EP + c];
      if (n0 + c < p.Cout) out[(size_t)which * p.Cout + n0 + c] = t;
    }
    if (kind == 1 && tid < CPR) {   // the shift (identical in all threads of a slot)
      const int n = n0 + slot * EP;
      if (n < p.Cout) {
#pragma unroll
        for (int e = 0; e < EP; ++e) out[(size_t)2 * p.Cout + n + e] = a0[e];
      }
    }
  }
};

__device__ __forceinline__ int xcd_remap(int bid, int n) {
  // give each XCD (blocks b, b+8, ... share one) a contiguous range of tiles: neighbours share halo rows in L2
  const int q = n >> 3, r = n & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}


static inline int pick_bn(int cout) { return cout > 64 ? 128 : (cout > 32 ? 64 : 32); }

// conv_patch.hip
bool conv_patch_applicable(const ConvParams& p, int dtype);
int conv_patch_stats_chunks(const ConvParams& p, int dtype);   // slab rows per sample of the column statistics, 0 = unsupported
int launch_conv_patch(const ConvParams& p, int dtype, hipStream_t s);
bool conv_patch_pair_applicable(const ConvParams& a, const ConvParams& b, int dtype);   // two convs of one shape in one launch
int launch_conv_patch_pair(const ConvParams& a, const ConvParams& b, hipStream_t s);
bool conv_tr2_applicable(const ConvParams& p, int dtype);   // stride-2 data gradient as four parity-class patch convs
int launch_conv_tr2(const ConvParams& p, hipStream_t s);
int conv_tr2_stats_chunks(const ConvParams& p, int dtype);   // slab rows per sample of its norm-backward sums, 0 = unsupported
#ifdef PWR_DEBUG_BUILD
void set_debug_stamps(long long* ptr);
void set_debug_delay(int d);
#endif

// conv_wstat.hip (round 5): the 128 -> 128 3x3 stride-1 bf16 conv with the weights stationary in registers, persistent workgroups;
// one job (b == nullptr) or two jobs of one geometry per launch
bool conv_wstat_applicable(const ConvParams& p, int dtype);
bool conv_wstat_narrow_applicable(const ConvParams& p, int dtype);   // 128 -> Cout <= 32, fp32 NCHW output (the heads' last conv)
bool conv_wstat_shape(int B, int H, int W, int Cin, int Cout, int ksize, int stride, int dtype);   // would a plain conv of this shape run on it?
bool conv_wstat_pair_applicable(const ConvParams& a, const ConvParams& b, int dtype);
bool conv_wstat_narrow_pair_applicable(const ConvParams& a, const ConvParams& b, int dtype);
int launch_conv_wstat(const ConvParams& a, const ConvParams* b, hipStream_t s);

// conv_wgrad_dma.hip: 3x3 weight gradient with both operands staged by LDS-DMA (operand already normalised: in_norm == null)
bool wgrad3d_applicable(const WgradParams& p);
int launch_wgrad3d(const WgradParams& p, hipStream_t s);

// conv_wgrad_ws.hip: 3x3 weight gradient of whole 128-channel tiles, wave-specialised (4 MFMA waves + 4 loader waves per workgroup);
// one job or two jobs of one geometry per launch
#ifdef PWR_DEBUG_BUILD
// tools/csrc_debug/conv_wgrad_ws9.hip (debug build only; measured slower): the same layers with all nine taps per workgroup
bool wgrad9w_applicable(const WgradParams& p);
int launch_wgrad9w(const WgradParams& p, hipStream_t s);
#endif
bool wgrad3w_applicable(const WgradParams& p);
int launch_wgrad3w(const WgradParams& a, const WgradParams* b, hipStream_t s);

}  // namespace pwr
