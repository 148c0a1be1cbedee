// Network engine: the whole PixelwiseRegression forward / backward as one static launch plan.
//
// The reference's forward (model.py:200-210) is ~350 ATen calls issued from Python per step; here the
// module's topology (stem -> stage x [1x1 -> hourglass -> plane/depth heads -> decoder]) is compiled
// once per (config, batch, dtype, training) into a list of kernel launches over a pre-planned arena:
// no allocation, no Python, no tracing compiler between kernels.  Activations stay resident in HBM
// for the backward pass (288 GB: nothing is recomputed except the fused norm+ReLU on operand load).
//
// Parameter order contract: the `params` table passed to pwr_engine_create lists (offset, numel) of
// every parameter in the reference's named_parameters() order; the builder
// consumes it in the same traversal and verifies every numel.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "pwr.h"
#include "pwr_common.h"
namespace pwr { bool conv_wstat_shape(int B, int H, int W, int Cin, int Cout, int ksize, int stride, int dtype); }   // conv_wstat.hip
#ifdef PWR_DEBUG_BUILD
#include "pwr_debug.h"
#endif

namespace {

struct Ctx {
  char* arena = nullptr;
  char* packs = nullptr;
  const float* params = nullptr;
  float* grads = nullptr;
  float* buffers = nullptr;  // BatchNorm running stats (flat)
  const float* img = nullptr;
  const float* label = nullptr;
  const float* mask = nullptr;
  float* out_p[8] = {};
  float* out_D[8] = {};
  float* out_uvd[8] = {};
  const float* g_p[8] = {};
  const float* g_D[8] = {};
  const float* g_uvd[8] = {};
  void* stream = nullptr;
  int training = 1;
  // second stream for the parameter-gradient kernels (wgrad + reduce + bias sums): they only feed the flat gradient
  // buffer, so they run beside the data-gradient / norm-backward chain instead of inside it
  // Several side streams, used round-robin, each with its own split-K slab: most of these kernels are small and latency-bound,
  // so independent layers' weight gradients overlap each other as well as the main chain.
  static constexpr int kMaxSide = 4;
  hipStream_t side[kMaxSide] = {};
  // Fork events: a pool used round-robin, one event per side-stream op in flight, instead of ONE event re-recorded for every op
  // (legal by the API -- a wait captures the record that precedes it -- but then every wait of a step hangs off one object).
  static constexpr int kForkPool = 64;
  hipEvent_t ev_fork[kForkPool] = {}, ev_join[kMaxSide] = {};
  int n_fork = 0, fork_rr = 0;
  int n_side = 0, side_rr = 0;
  size_t slab_off = 0, slab_stride = 0;   // byte offset of the current stream's slab inside the slab scratch
  size_t napply_off = 0, napply_stride = 0;   // the same for the materialised-operand scratch (Engine::scr_napply)
  bool use_side = true;
  bool attached = false;    // side streams / events taken from the process-wide pool
  // Side ops held back while `defer` is set (run_on_side appends them here), issued by the op that clears it: the heads' weight gradients
  // wait until the heads' data-gradient chain -- full-chip kernels -- has been issued and then run beside the hourglass backward, whose
  // kernels are small (Engine::heads_bwd)
  struct Deferred { std::function<int(Ctx&)> fn; bool light; };
  std::vector<Deferred> deferred;
  bool defer = false;
};

// The side streams, their join events and the pool of fork events are PROCESS-wide (one set per device), shared by every engine:
// HIP streams map onto a limited number of hardware queues, and every further pair of side streams kept alive by another plan (a
// second batch size, the previous model of a sweep not yet garbage-collected) made ALL of them slower -- C3's train step took
// 20.2 ms instead of 11.5 ms when it ran after C2's plan in the same process.  The streams live as long as the process.
// Threading contract (as in the reference: a single-threaded caller): the pool's creation is locked, but the
// fork / join events are shared, so backward calls of different engines on the same device must not run concurrently.
struct SidePool {
  hipStream_t side[Ctx::kMaxSide] = {};
  hipEvent_t ev_fork[Ctx::kForkPool] = {}, ev_join[Ctx::kMaxSide] = {};
  int n_side = -1, n_fork = 0;     // -1: not created yet
};
static std::mutex g_side_mu;
static void side_pool_attach(Ctx& c, hipStream_t caller) {
  // keyed by the device the CALLER's stream lives on (not the thread's current device), created once under a lock: two host
  // threads may enter pwr_engine_backward together (ctypes drops the GIL)
  std::lock_guard<std::mutex> lk(g_side_mu);
  static SidePool pools[16];
  int dev = 0, cur = 0;
  (void)hipGetDevice(&cur);
  if (hipStreamGetDevice(caller, &dev) != hipSuccess) { (void)hipGetLastError(); dev = cur; }
  SidePool& sp = pools[dev & 15];
  if (sp.n_side < 0) {
    if (dev != cur) (void)hipSetDevice(dev);
    int want = PWR_DBG_ENV("PWR_SIDE_STREAM", 2);   // number of side streams, 0 = everything on the caller's stream
    if (want > Ctx::kMaxSide) want = Ctx::kMaxSide;
    sp.n_side = 0;
    for (int k = 0; want > 0 && k < Ctx::kForkPool; ++k) {
      if (hipEventCreateWithFlags(&sp.ev_fork[k], hipEventDisableTiming) != hipSuccess) break;
      sp.n_fork = k + 1;
    }
    // side streams at the LOWEST priority: when a weight-gradient kernel and a kernel of the critical chain both have
    // workgroups to place, the chain goes first and the weight gradients fill what is left
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    const bool low = PWR_DBG_ENV("PWR_SIDE_PRIORITY", 1) != 0;
    // DEBUG build: side streams confined to PWR_SIDE_CUS compute units (hipExtStreamCreateWithCUMask; pattern 0 = the first n mask
    // bits, 1 = every (256/n)-th bit, 2 = the LAST n bits) -- the CU-partitioning experiment of tools/step_elimination.py
    const int side_cus = PWR_DBG_ENV("PWR_SIDE_CUS", 0), side_pat = PWR_DBG_ENV("PWR_SIDE_CU_PATTERN", 0);
    for (int k = 0; sp.n_fork > 0 && k < want; ++k) {
      hipError_t er;
      if (side_cus > 0 && side_cus < 256) {
        uint32_t mask[8] = {};
        for (int i = 0; i < side_cus; ++i) {
          const int bit = side_pat == 1 ? i * (256 / side_cus) : (side_pat == 2 ? 255 - i : i);
          mask[bit >> 5] |= 1u << (bit & 31);
        }
        er = hipExtStreamCreateWithCUMask(&sp.side[k], 8, mask);
      } else {
        er = hipStreamCreateWithPriority(&sp.side[k], hipStreamNonBlocking, low ? prio_lo : 0);
      }
      if (er != hipSuccess || hipEventCreateWithFlags(&sp.ev_join[k], hipEventDisableTiming) != hipSuccess) break;
      sp.n_side = k + 1;
    }
    if (dev != cur) (void)hipSetDevice(cur);
  }
  for (int k = 0; k < sp.n_side; ++k) { c.side[k] = sp.side[k]; c.ev_join[k] = sp.ev_join[k]; }
  for (int k = 0; k < sp.n_fork; ++k) c.ev_fork[k] = sp.ev_fork[k];
  c.n_side = sp.n_side; c.n_fork = sp.n_fork;
  c.use_side = sp.n_side > 0;
  c.attached = true;
}

// DEBUG build only -- step-level timing by elimination (tools/step_elimination.py; the results are WRONG by construction):
// bit 0: no parameter-gradient launches (everything that goes to the side streams), bit 1: no norm-backward launches,
// bit 2: no data-gradient convs / fused ResBlock backwards, bit 3: no fork events (side launches unordered).  Constant 0 in the
// shipped library.
static int elim_mask() {
  static const int m = PWR_DBG_ENV("PWR_ELIM", 0);
  return m;
}

// run `op` on a side stream (round-robin), ordered after everything enqueued so far on the caller's stream.
// One fork event per op.  Measured in round 3 (profiles/r3_step_elimination.json): with the side launches NOT ordered behind the chain
// the step takes 0.33 ms less -- but batching several ops behind one event (fewer markers in the chain's queue, each op issued a few
// chain kernels later) made the step SLOWER, 6.47 ms at one op per event against 6.8 - 7.1 ms at 2 .. 12: what counts is that the
// parameter-gradient work starts as early as it can, because the chain waits for it at the end of every segment.
// light: a launch of a few microseconds (the parameter sums of a norm backward, round 6) goes BEHIND the op issued last -- onto that op's
// stream -- and does not advance the round-robin: in the hourglass / stem sequences weight gradients and norm backwards alternate, so with
// two side streams a light op that took its own turn put EVERY weight gradient on one stream (measured: +0.25 ms per step).
static inline int run_on_side(Ctx& c, const std::function<int(Ctx&)>& op, bool forked = false, bool light = false) {
  if (elim_mask() & 1) return 0;
  if (!c.use_side || c.n_side == 0) return op(c);
  if (c.defer) { c.deferred.push_back(Ctx::Deferred{op, light}); return 0; }
  const int k = light ? (c.side_rr + c.n_side - 1) % c.n_side : c.side_rr;
  if (!light) c.side_rr = (k + 1) % c.n_side;
  if (!forked && !(elim_mask() & 8)) {     // (bit 3, debug build: side ops NOT ordered behind the chain; forked: the caller has ordered the side streams behind the chain already)
    hipEvent_t ev = c.ev_fork[c.fork_rr];
    c.fork_rr = (c.fork_rr + 1) % c.n_fork;
    hipEventRecord(ev, (hipStream_t)c.stream);
    hipStreamWaitEvent(c.side[k], ev, 0);
  }
  void* main_stream = c.stream;
  c.stream = c.side[k];
  c.slab_off = (size_t)k * c.slab_stride;
  c.napply_off = (size_t)k * c.napply_stride;
  const int rc = op(c);
  c.stream = main_stream;
  c.slab_off = 0;
  c.napply_off = 0;
  return rc;
}

// A launch op, tagged with the scope (network part: "stem", "s0.hg3", "s1.heads", "s1.plane.bwd" ...) that was current when it was built --
// the backward lists are built in reverse and spliced, so the tag travels with the op.  Used by the debug build's per-scope timing
// (pwr_engine_set_timing / pwr_engine_timing_report, include/pwr_debug.h); free otherwise.
// The tags exist in the DEBUG build only.  There the current tag is a thread-local pointer into strings owned by the Engine being built
// (ScopeName::names, freed with the plan): two plans built concurrently on different threads do not see each other's tags, and nothing
// outlives its plan.  The product build carries no tag, allocates nothing per scope change and has no global to race on.
#ifdef PWR_DEBUG_BUILD
static thread_local const char* g_scope_tag = "";
#endif
struct ScopeName {
  std::string s;
#ifdef PWR_DEBUG_BUILD
  std::vector<std::unique_ptr<std::string>> names;
#endif
  ScopeName& operator=(const std::string& v) {
    s = v;
#ifdef PWR_DEBUG_BUILD
    names.emplace_back(new std::string(v));
    g_scope_tag = names.back()->c_str();
#endif
    return *this;
  }
};
struct Op {
  std::function<int(Ctx&)> f;
  const char* tag;
  Op() : tag("") {}
  template <class F, class = typename std::enable_if<!std::is_same<typename std::decay<F>::type, Op>::value>::type>
#ifdef PWR_DEBUG_BUILD
  Op(F&& fn) : f(std::forward<F>(fn)), tag(g_scope_tag) {}
#else
  Op(F&& fn) : f(std::forward<F>(fn)), tag("") {}
#endif
  int operator()(Ctx& c) const { return f(c); }
};

struct Tn {  // NHWC activation in the arena
  size_t off = 0, goff = 0;
  int H = 0, W = 0, C = 0;
};
struct NormL {
  size_t sums = 0;  // [B][2][C] fp32 per-sample backward sums (small maps)
  size_t state;  // [4][B][C] fp32: mean, rstd, scale, beta
  long long gamma, beta, rm = -1, rv = -1;
  int C;
};
struct ConvL {
  long long w, b;
  size_t pack_f = 0, pack_d = 0;
  int Cin, Cout, k, stride;
  int cin_real = 0;  // < Cin when the input tensor carries zero-padded channels (stage-input concat)
};
struct PackDescHost {
  long long src_off, dst_off;
  int Cout, Cin, ksize, kind, rows_pad, KCH, dtype, pad_;
};

struct Engine {
  // config
  int J, stages, P, F, level, ks, norm_mode, method;
  int B, dtype, training;
  int esz;
  // parameter table
  std::vector<long long> poff, pnum;
  size_t pcur = 0;
  std::vector<long long> boff;  // running stat buffers (rm, rv per norm) offsets
  size_t bcur = 0;
  // plan
  size_t arena_bytes = 0, pack_bytes = 0;
  std::vector<Op> fwd;
  std::vector<std::vector<Op>> bwd;  // per segment, already in execution order
  std::vector<Op> bwd_cur;           // segment being built (reverse order)
  std::vector<PackDescHost> descs;
  size_t desc_dev_off = 0;  // descs live at the start of the pack buffer
  // shared scratch
  size_t scr_partial = 0, scr_S1 = 0, scr_S2 = 0, scr_slab = 0, scr_slab_bytes = 0;
  size_t need_partial = 0, need_sc = 0;
  size_t scr_cpartial = 0, need_cpartial = 0;   // column statistics written by conv epilogues (pwr_conv_fwd_stats)
  // a tensor whose producer (max-pool / up-sample + add) was left to the one-launch ResBlock that consumes it (resblock_fused)
  struct LazyX { bool valid = false; int mode = 0; size_t x_off = 0, a_off = 0, h_off = 0; };
  LazyX lazy_x;
  // the same for the backward pass (round 4): the gradient of the up-sample is summed by the one-launch ResBlock backward that consumes it
  // while it loads (pend_up: the level's `out` gradient), the max-pool's gradient is routed by the one that produces it while it stores
  // (pend_pool: x_off = the pooled tensor the block must read, a / addend / dst = the level's input-block output, its `out` gradient and
  // the gradient buffer of that output); hourglass() registers them, resblock_fused() consumes them
  struct PendUp { bool valid = false; size_t src_goff = 0; };
  struct PendPool { bool valid = false; size_t x_off = 0, a_off = 0, addend_goff = 0, dst_goff = 0; };
  PendUp pend_up;
  PendPool pend_pool;
  int hg_depth = 0;
  // The ONE producer/consumer pair of such statistics that spans two backward segments (two C ABI calls, with the all-reduce and
  // Python in between): stage 0's input-conv data gradient writes the norm-backward reductions of the stem's last norm, the stem
  // segment's first op reads them.  It has a buffer of its own so that nothing else can ever touch it in between.
  size_t scr_handoff = 0, need_handoff = 0;
  // round 6: relu(norm(.)) of a head tensor materialised for its weight gradient (pwr_norm_apply): one buffer per side stream, like the slabs
  size_t scr_napply = 0, need_napply = 0;
  std::string err;
  long long generation = 0;
  // round 6: the weight re-pack of a forward runs on a side stream beside the stem's first conv (Cin = 1: VALU, reads the fp32 parameters) and
  // its norm statistics; the chain waits for it in front of the first op that reads a pack (pwr_engine_pack_beside_forward)
  bool pack_pending = false;
  size_t fwd_first_pack_op = ~(size_t)0;
  hipEvent_t ev_pack = nullptr;
  Ctx ctx;
  bool join_each_segment = PWR_DBG_ENV("PWR_JOIN_EACH", 0) != 0;      // (debug build: PWR_JOIN_EACH=1 makes the caller's stream wait for the side streams after EVERY segment, rounds 2 - 3)

  // arena layout record (debugging aid: pwr_engine_layout)
  struct AllocRec { size_t off, bytes; std::string tag; };
  std::vector<AllocRec> layout;
  ScopeName scope;
  int cur_stage = -1;
#ifdef PWR_DEBUG_BUILD
  // per-scope timing of the CHAIN (the caller's stream): an event at every change of scope tag in the op sequence of a forward / a backward
  // segment; pwr_engine_timing_report synchronises and adds up.  (The side streams' weight-gradient kernels are not in these numbers.)
  bool timing = false;
  struct TimedEv { std::string phase; const char* tag; hipEvent_t ev; };
  std::vector<TimedEv> timed;
  void mark(const std::string& phase, const char* tag, void* st) {
    hipEvent_t ev;
    hipEventCreate(&ev);
    hipEventRecord(ev, (hipStream_t)st);
    timed.push_back({phase, tag, ev});
  }
#endif
  // Bump allocation, once per plan: every activation / gradient buffer has its own arena range for the life of the plan (nothing is aliased
  // across layers or backward segments).  The single join of the side streams after the LAST backward segment (pwr_engine_backward) relies
  // on exactly that: a chain op of a later segment can never overwrite what a side-stream op of an earlier one still reads.
  size_t alloc(size_t bytes, const char* tag = "") {
    size_t o = arena_bytes;
    arena_bytes += (bytes + 255) / 256 * 256;
    layout.push_back({o, bytes, scope.s + ":" + tag + "#" + std::to_string(layout.size())});
    return o;
  }
  size_t alloc_pack(size_t bytes) {
    size_t o = pack_bytes;
    pack_bytes += (bytes + 255) / 256 * 256;
    return o;
  }
  Tn tensor(int H, int W, int C, bool grad) {
    Tn t;
    t.H = H; t.W = W; t.C = C;
    char tg[64];
    snprintf(tg, sizeof tg, "act%dx%dx%d", H, W, C);
    t.off = alloc((size_t)B * H * W * C * esz, tg);
    tg[0] = 'g'; tg[1] = 'r'; tg[2] = 'd';
    if (grad) t.goff = alloc((size_t)B * H * W * C * esz, tg);
    return t;
  }
  long long take_param(long long numel) {
    if (pcur >= poff.size()) { err = "parameter table too short"; return 0; }
    if (pnum[pcur] != numel) {
      char b[128];
      snprintf(b, sizeof b, "parameter %zu: expected %lld elements, table has %lld", pcur, numel, pnum[pcur]);
      err = b;
    }
    return poff[pcur++];
  }
  long long take_buffer() {
    if (bcur >= boff.size()) { err = "buffer table too short"; return 0; }
    return boff[bcur++];
  }
  void want_slab(size_t bytes) { if (bytes > scr_slab_bytes) scr_slab_bytes = bytes; }

  // ---------------------------------------------------------------- layer constructors
  // map_hw > 0: the layer runs on map_hw x map_hw maps; where conv_wstat.hip takes that shape its packs are written in that kernel's fragment
  // order and handed over with bit 0 of the address set (conv_mfma.hip PackDesc::order)
  ConvL conv_params(int cin, int cout, int k, int stride, bool mfma, bool need_dgrad, int map_hw = 0) {
    ConvL c;
    c.Cin = cin; c.Cout = cout; c.k = k; c.stride = stride; c.cin_real = cin;
    c.w = take_param((long long)cout * cin * k * k);
    c.b = take_param(cout);
    if (mfma) {
      PackDescHost d;
      const int KE = dtype == PWR_BF16 ? 32 : 16;
      const bool frag = map_hw > 0 && pwr::conv_wstat_shape(B, map_hw, map_hw, cin, cout, k, stride, dtype);
      c.pack_f = alloc_pack(pwr_conv_pack_bytes(cout, cin, k, 0, dtype));
      d = {c.w, (long long)c.pack_f, cout, cin, k, 0, pwr_conv_out_pad(cout), (cin + KE - 1) / KE, dtype, frag ? 1 : 0};
      descs.push_back(d);
      if (frag) c.pack_f |= 1;
      if (need_dgrad && training) {
        const int kind = stride == 2 ? 2 : 1;
        c.pack_d = alloc_pack(pwr_conv_pack_bytes(cout, cin, k, kind, dtype));
        d = {c.w, (long long)c.pack_d, cout, cin, k, kind, pwr_conv_out_pad(cin), (cout + KE - 1) / KE, dtype, 0};
        descs.push_back(d);
      }
    }
    return c;
  }
  NormL norm_params(int C) {
    NormL n;
    n.C = C;
    n.gamma = take_param(C);
    n.beta = take_param(C);
    if (norm_mode == 1) { n.rm = take_buffer(); n.rv = take_buffer(); }
    n.state = alloc((size_t)4 * B * C * 4, "nstate");
    if (training) n.sums = alloc((size_t)2 * B * C * 4, "nsums");
    return n;   // (the backward partial slab is sized per tensor in norm_bwd)
  }
  int splits_for(int M, int cin, int cout, int k, int stride = 1) const {
    const int KE = dtype == PWR_BF16 ? 32 : 16;
    const int steps = (M + KE - 1) / KE;
    const int per = ((cin + 127) / 128) * (pwr_conv_out_pad(cout) / (cout > 64 ? 128 : (cout > 32 ? 64 : 32)));
    // bf16 3x3 stride-1 layers on W % 32 == 0 maps use the 3-taps-per-workgroup kernel (grid 3 x tiles x splits,
    // one workgroup per CU); everything else one tap per workgroup at >= 2 workgroups per CU
    const bool w3 = dtype == PWR_BF16 && k == 3 && M % 32 == 0;
    const int tiles = (w3 ? 3 : k * k) * per;
    static const int w3_target = PWR_DBG_ENV("PWR_WGRAD3_SLOTS", 256);
    // w3: one wave of workgroups, whole XCD groups (no tail).  The <= 64-channel kernels fit two workgroups per CU; on the big
    // stem maps (>= 2^18 pixels: 205 K steps per workgroup otherwise, at the tail of the step) they get two waves of them.
    // (PWR_STEM_SPLITS2X=1, debug build: also the stem's 64 -> 128 layer -- its 64 x 128 tile holds 96 accumulators, two workgroups fit a
    // CU and 80 splits are ONE per CU: 180 us alone against 117 at 160 splits, and NOTHING in the train step, 5.57 ms either way: at the
    // tail of the step the kernel shares the chip with the stem's data gradients whatever its split count)
    const bool two_rounds = PWR_DBG_ENV("PWR_STEM_SPLITS2X", 0) ? (cin <= 64 || cout <= 64) : cout <= 64;
    const int target = (w3 && two_rounds && M >= (1 << 18)) ? 2 * w3_target : w3_target;
    static const int tr_target = PWR_DBG_ENV("PWR_WGRAD_TR_SLOTS", 512);
    int s = w3 ? (target / tiles) / 8 * 8 : (tr_target + tiles - 1) / tiles;
    if (w3 && s < 8) s = 8;
    // whole 128-channel tiles on both sides: the wave-specialised kernel (conv_wgrad_ws.hip).  Its 512-thread workgroup takes a whole CU
    // (8 waves x 250 registers), so it is given FEW, long-running workgroups -- 72 = 24 splits x 3 kernel rows of a 128 -> 128 layer --
    // and the other CUs stay free for the chain: measured on the train step (same box, profiles/r4_experiments.md): 80 splits (240
    // workgroups: every small chain kernel waits for a CU) 6.28 ms, 48: 6.00, 32: 5.91, 24: 5.82 - 5.87, 16: 5.97, 12: 6.45; the
    // register-staged kernel at its 80 splits: 6.01 - 6.14.  At least one split per six samples (the kernel keeps <= 8 samples' norm states).
    const int ws_on = PWR_DBG_ENV("PWR_WGRAD3W", 1);
    // (stride 2 -- the stem's last conv -- runs the three-tap kernel's stride-2 form at the w3 split count: 2 x 3 x 80 workgroups)
    if (w3 && stride == 1 && (cin % 128 == 0 || (cin == 64 && ws_on == 3)) && cout % 128 == 0 && ws_on) {
      // (nine-tap form: (cin / 64) x (cout / 64) workgroups per split; three-tap form, debug build: 3 x per)
      s = PWR_DBG_ENV("PWR_WGRAD9W", 0) ? PWR_DBG_ENV("PWR_WGRAD9W_WGS", 72) / (4 * per) : PWR_DBG_ENV("PWR_WGRAD3W_SPLITS", 24) / per;
      // (the stem's 64 -> 128 layer at 128 x 128: four times the K extent of a head layer at half the work per K step: twice the workgroups)
      if (cin == 64) s = PWR_DBG_ENV("PWR_WGRAD3W_SPLITS64", 48) / per;
      if (s < (B + 5) / 6) s = (B + 5) / 6;
      if (s < 1) s = 1;
    }
    const int maxs = steps / 8 > 0 ? steps / 8 : 1;
    if (s > maxs) s = maxs;
    if (s < 1) s = 1;
    return s;
  }

  // A/B switch: bit 0 = forward statistics from the conv epilogues, bit 1 = norm-backward reductions from the data-gradient epilogues
  static int stats_mask() {
    static const int m = PWR_DBG_ENV("PWR_CONV_STATS_MASK", 3);
    return m;
  }
  // ---- norm statistics of tensor t (forward) and its backward (g -> dy, in place in t.goff, + addend)
  void norm_fwd_sizes(const Tn& t) {
    const size_t pb = pwr_norm_partial_bytes(B, t.H * t.W, t.C);
    if (pb > need_partial) need_partial = pb;
    if ((size_t)B * t.C * 4 > need_sc) need_sc = (size_t)B * t.C * 4;
  }
  void norm_fwd(const Tn& t, const NormL& n) {
    const int HW = t.H * t.W, C = t.C, Bc = B, dt = dtype;
    norm_fwd_sizes(t);
    const int nm = norm_mode;
    Engine* E = this;
    fwd.push_back([=](Ctx& c) {
      int mode = nm == 0 ? 0 : (c.training ? 1 : 2);
      float* rm = n.rm >= 0 ? c.buffers + n.rm : nullptr;
      float* rv = n.rv >= 0 ? c.buffers + n.rv : nullptr;
      return pwr_norm_stats(c.arena + t.off, c.params + n.gamma, c.params + n.beta, rm, rv,
                            (float*)(c.arena + E->scr_partial), (float*)(c.arena + n.state), Bc, HW, C, mode, 1e-5f, 0.1f, dt,
                            c.stream);
    });
  }
  // round 6: the statistics launch of a ResBlock's first norm with the tensor's producer fused in (hourglass(): the level's max-pool, the
  // inner level's up-sample + skip, left pending in lazy_x for an UNFUSED ResBlock): two launches instead of three, same bytes
  static bool src_fuse_on() {
    static const bool on = PWR_DBG_ENV("PWR_NORM_SRC_FUSE", 1) != 0;
    return on;
  }
  void norm_fwd_src(const Tn& t, const NormL& n, const LazyX& lx) {
    const int C = t.C, Bc = B, dt = dtype, Hc = t.H, Wc = t.W;
    norm_fwd_sizes(t);
    Engine* E = this;
    fwd.push_back([=](Ctx& c) {
      int rc = pwr_norm_stats_fused_src(lx.mode, c.arena + lx.a_off, lx.mode == 2 ? c.arena + lx.h_off : nullptr, c.arena + t.off, c.params + n.gamma,
                                        c.params + n.beta, (float*)(c.arena + E->scr_partial), (float*)(c.arena + n.state), Bc, Hc, Wc, C, 1e-5f, dt, c.stream);
      if (rc != PWR_EUNSUPPORTED) return rc;
      rc = lx.mode == 1 ? pwr_maxpool_fwd(c.arena + lx.a_off, c.arena + t.off, Bc, 2 * Hc, 2 * Wc, C, dt, c.stream)
                        : pwr_upsample_add_fwd(c.arena + lx.h_off, c.arena + lx.a_off, c.arena + t.off, Bc, Hc / 2, Wc / 2, Hc, Wc, C, dt, c.stream);
      if (rc) return rc;
      return pwr_norm_stats(c.arena + t.off, c.params + n.gamma, c.params + n.beta, nullptr, nullptr, (float*)(c.arena + E->scr_partial),
                            (float*)(c.arena + n.state), Bc, Hc * Wc, C, 0, 1e-5f, 0.1f, dt, c.stream);
    });
  }
  // grad buffer of t holds g = dL/d relu(norm(t)); result dy replaces it (plus addend tensor's grad if addend_goff != 0)
  // chunks > 0: the data-gradient conv that produced g already wrote the two reductions (conv_bwd's return value)
  // Round 6 (fold): with instance norm and a slab of its OWN (own_off: conv_bwd allocated it for this layer; or the hand-off buffer), the
  // reduction launch leaves the chain: ONE apply launch that sums the slab rows of its sample itself (pwr_norm_bwd_apply_from_partial,
  // bit-identical dy) and the parameter sums on a side stream from the same slab (nothing else ever writes it within a step).
  static constexpr size_t kNoOwn = ~(size_t)0;
  // dgamma / dbeta of the folded norm backwards of the segment being built: ONE grouped launch on a side stream at the segment's end
  // (pwr_norm_bwd_params_group; as a launch per layer they were 28 more small kernels per step on the side streams, and the step -- which
  // ends when the side streams do -- got longer than with the reduction on the chain)
  struct NormJob { size_t off; bool handoff; long long gamma, beta; int HW, C, chunks; };
  std::vector<NormJob> seg_norm_jobs;
  void flush_norm_jobs() {
    if (seg_norm_jobs.empty()) return;
    const std::vector<NormJob> jobs = seg_norm_jobs;
    seg_norm_jobs.clear();
    const int Bc = B;
    Engine* E = this;
    bwd_cur.push_back([=](Ctx& c) {
      if (elim_mask() & 2) return 0;
      return run_on_side(c, [=](Ctx& c2) {
        std::vector<pwr_norm_param_job> arr;
        for (auto& j : jobs)
          arr.push_back(pwr_norm_param_job{(const float*)(c2.arena + (j.handoff ? E->scr_handoff + j.off : j.off)), c2.grads + j.gamma, c2.grads + j.beta,
                                           j.HW, j.C, j.chunks});
        return pwr_norm_bwd_params_group(arr.data(), (int)arr.size(), 0, Bc, c2.stream);
      }, false, true);
    });
  }
  bool fold_ok(int C) const {
    static const bool on = PWR_DBG_ENV("PWR_NORM_BWD_FOLD", 1) != 0;
    return on && norm_mode == 0 && 2 * C <= 1024;
  }
  void norm_bwd(const Tn& t, const NormL& n, size_t addend_goff, bool has_addend, int chunks = 0, bool handoff = false, size_t cpart_off = 0,
                size_t own_off = kNoOwn) {
    const int HW = t.H * t.W, C = t.C, Bc = B, dt = dtype, nm = norm_mode;
    Engine* E = this;
    const size_t Engine::*cpart = handoff ? &Engine::scr_handoff : &Engine::scr_cpartial;
    if ((size_t)B * C * 4 > need_sc) need_sc = (size_t)B * C * 4;
    const bool own = own_off != kNoOwn;
    const bool fold = chunks > 0 && (own || handoff) && fold_ok(C);
    if (fold) seg_norm_jobs.push_back(NormJob{own ? own_off : cpart_off, !own, n.gamma, n.beta, HW, C, chunks});
    // (Measured and dropped, DESIGN.md section 4: one-block-per-sample, split and deferred-dgamma forms of this step.)
    bwd_cur.push_back([=](Ctx& c) {
      if (elim_mask() & 2) return 0;
      const int mode = nm == 0 ? 0 : (c.training ? 1 : 2);
      if (chunks > 0 && mode != 2) {  // (eval-mode batch norm: statistics are constants, the plain path handles it)
        float* part = own ? (float*)(c.arena + own_off) : (float*)(c.arena + E->*cpart + cpart_off);
        if (fold) {
          int rc = pwr_norm_bwd_apply_from_partial(c.arena + t.goff, c.arena + t.off, (float*)(c.arena + n.state), part,
                                                   has_addend ? c.arena + addend_goff : nullptr, c.arena + t.goff, nullptr, nullptr, nullptr, nullptr,
                                                   nullptr, chunks, 1, Bc, HW, C, dt, c.stream);
          return rc;     // (dgamma / dbeta: the segment's grouped launch, flush_norm_jobs)
        }
        return pwr_norm_bwd_from_partial(c.arena + t.goff, c.arena + t.off, (float*)(c.arena + n.state), part,
                                         chunks, (float*)(c.arena + E->scr_S1), (float*)(c.arena + E->scr_S2),
                                         has_addend ? c.arena + addend_goff : nullptr, c.arena + t.goff, c.grads + n.gamma,
                                         c.grads + n.beta, 0, 1, Bc, HW, C, mode, dt, c.stream);
      }
      return pwr_norm_bwd(c.arena + t.goff, c.arena + t.off, (float*)(c.arena + n.state), (float*)(c.arena + E->scr_partial),
                          (float*)(c.arena + E->scr_S1), (float*)(c.arena + E->scr_S2),
                          has_addend ? c.arena + addend_goff : nullptr, c.arena + t.goff, c.grads + n.gamma, c.grads + n.beta, 0,
                          1, Bc, HW, C, mode, dt, c.stream);
    });
  }

  // the norm backwards of the two heads' tensors of one depth (instance norm, reductions already written by the paired data gradient at
  // cpartial + 0 / + off_b): two launches instead of four on a stretch where the chain runs alone
  // own_off != kNoOwn: the two slabs live at own_off / own_off + off_b in a buffer of this depth's own -> the fold (see norm_bwd)
  void norm_bwd_pair(const Tn& ta, const NormL& na, const Tn& tb, const NormL& nb, int chunks, size_t off_b, size_t own_off = kNoOwn) {
    static const bool on = PWR_DBG_ENV("PWR_NORM_BWD_PAIR", 1) != 0;
    if (!on || norm_mode != 0 || chunks <= 0 || ta.H != tb.H || ta.W != tb.W || ta.C != tb.C) {
      norm_bwd(ta, na, 0, false, chunks, false, 0, own_off);
      norm_bwd(tb, nb, 0, false, chunks, false, off_b, own_off == kNoOwn ? kNoOwn : own_off + off_b);
      return;
    }
    const int HW = ta.H * ta.W, C = ta.C, Bc = B, dt = dtype;
    Engine* E = this;
    if ((size_t)2 * B * C * 4 > need_sc) need_sc = (size_t)2 * B * C * 4;
    const bool own = own_off != kNoOwn;
    const bool fold = own && fold_ok(C);
    if (fold) {
      seg_norm_jobs.push_back(NormJob{own_off, false, na.gamma, na.beta, HW, C, chunks});
      seg_norm_jobs.push_back(NormJob{own_off + off_b, false, nb.gamma, nb.beta, HW, C, chunks});
    }
    bwd_cur.push_back([=](Ctx& c) {
      if (elim_mask() & 2) return 0;
      float* pa = own ? (float*)(c.arena + own_off) : (float*)(c.arena + E->scr_cpartial);
      float* pb = own ? (float*)(c.arena + own_off + off_b) : (float*)(c.arena + E->scr_cpartial + off_b);
      if (fold) {
        int rc = pwr_norm_bwd_apply_from_partial(c.arena + ta.goff, c.arena + ta.off, (float*)(c.arena + na.state), pa, nullptr, c.arena + ta.goff,
                                                 c.arena + tb.goff, c.arena + tb.off, (float*)(c.arena + nb.state), pb, c.arena + tb.goff, chunks, 1, Bc,
                                                 HW, C, dt, c.stream);
        return rc;     // (dgamma / dbeta: the segment's grouped launch, flush_norm_jobs)
      }
      return pwr_norm_bwd_from_partial_pair(c.arena + ta.goff, c.arena + ta.off, (float*)(c.arena + na.state), pa,
                                            c.arena + ta.goff, c.grads + na.gamma, c.grads + na.beta, c.arena + tb.goff, c.arena + tb.off,
                                            (float*)(c.arena + nb.state), pb, c.arena + tb.goff,
                                            c.grads + nb.gamma, c.grads + nb.beta, chunks, (float*)(c.arena + E->scr_S1),
                                            (float*)(c.arena + E->scr_S2), 0, 1, Bc, HW, C, dt, c.stream);
    });
  }

  // ---- MFMA conv: y = conv(NR(x)) + bias (+ residual)
  // out_norm: the norm that follows the conv in model.py; its statistics come out of the conv's epilogue when the shape
  // allows (otherwise conv, then the standalone statistics kernels)
  Tn conv_fwd(const Tn& x, const NormL* nr, const ConvL& cv, const Tn* residual, bool grad, const NormL* out_norm = nullptr) {
    const int pad = cv.k / 2;
    const int Ho = (x.H + 2 * pad - cv.k) / cv.stride + 1, Wo = (x.W + 2 * pad - cv.k) / cv.stride + 1;
    Tn y = tensor(Ho, Wo, cv.Cout, grad);
    const int Bc = B, dt = dtype;
    const bool has_nr = nr != nullptr;
    const NormL n = has_nr ? *nr : NormL{};
    const bool has_res = residual != nullptr;
    const size_t roff = has_res ? residual->off : 0;
    const int chunks = (out_norm && (stats_mask() & 1)) ? pwr_conv_stats_chunks(x.H, x.W, cv.Cin, cv.Cout, cv.k, cv.stride, 0, dtype) : 0;
    if (chunks > 0) {
      const NormL on = *out_norm;
      const size_t need = (size_t)B * chunks * 3 * cv.Cout * 4;
      if (need > need_cpartial) need_cpartial = need;
      const int nm = norm_mode, HWo = Ho * Wo;
      Engine* E = this;
      norm_fwd_sizes(y);
      fwd.push_back([=](Ctx& c) {
        const int mode = nm == 0 ? 0 : (c.training ? 1 : 2);
        float* rm = on.rm >= 0 ? c.buffers + on.rm : nullptr;
        float* rv = on.rv >= 0 ? c.buffers + on.rv : nullptr;
        if (mode == 2) {
          int rc = pwr_conv_fwd(c.arena + x.off, c.packs + cv.pack_f, c.params + cv.b, has_nr ? (float*)(c.arena + n.state) : nullptr,
                                1, has_res ? c.arena + roff : nullptr, c.arena + y.off, nullptr, Bc, x.H, x.W,
                                cv.Cin, cv.Cout, cv.k, cv.stride, 0, dt, c.stream);
          if (rc) return rc;
          return pwr_norm_stats(c.arena + y.off, c.params + on.gamma, c.params + on.beta, rm, rv, (float*)(c.arena + E->scr_partial),
                                (float*)(c.arena + on.state), Bc, HWo, cv.Cout, mode, 1e-5f, 0.1f, dt, c.stream);
        }
        int rc = pwr_conv_fwd_stats(c.arena + x.off, c.packs + cv.pack_f, c.params + cv.b, has_nr ? (float*)(c.arena + n.state) : nullptr,
                                    1, has_res ? c.arena + roff : nullptr, c.arena + y.off, Bc, x.H, x.W, cv.Cin,
                                    cv.Cout, cv.k, cv.stride, 0, (float*)(c.arena + E->scr_cpartial), nullptr, nullptr, nullptr, 1, dt, c.stream);
        if (rc) return rc;
        return pwr_norm_finalize_partial((float*)(c.arena + E->scr_cpartial), chunks, c.params + on.gamma,
                                         c.params + on.beta, rm, rv, (float*)(c.arena + on.state), Bc, HWo, cv.Cout, mode, 1e-5f, 0.1f,
                                         c.stream);
      });
      return y;
    }
    fwd.push_back([=](Ctx& c) {
      return pwr_conv_fwd(c.arena + x.off, c.packs + cv.pack_f, c.params + cv.b, has_nr ? (float*)(c.arena + n.state) : nullptr,
                          1, has_res ? c.arena + roff : nullptr, c.arena + y.off,
                          nullptr, Bc, x.H, x.W, cv.Cin, cv.Cout, cv.k, cv.stride, 0, dt, c.stream);
    });
    if (out_norm) norm_fwd(y, *out_norm);
    return y;
  }
  // Two convs of the same shape on different tensors / weights, each followed by its norm (the two regression heads of a stage): one
  // launch for both when the shape allows (pwr_conv_fwd_stats_pair), then the two finalize launches.  Forward only: the backward ops
  // are registered per head as before.
  void conv_fwd_pair(const Tn& xa, const NormL* nra, const ConvL& ca, const NormL& ona, Tn& ya,
                     const Tn& xb, const NormL* nrb, const ConvL& cb, const NormL& onb, Tn& yb, bool grad) {
    static const bool on = PWR_DBG_ENV("PWR_HEAD_PAIR", 1) != 0;
    const int chunks = (stats_mask() & 1) ? pwr_conv_stats_chunks(xa.H, xa.W, ca.Cin, ca.Cout, ca.k, ca.stride, 0, dtype) : 0;
    const bool same = xa.H == xb.H && xa.W == xb.W && ca.Cin == cb.Cin && ca.Cout == cb.Cout && ca.k == cb.k && ca.stride == 1 && cb.stride == 1;
    if (!on || chunks <= 0 || !same || norm_mode != 0) {
      ya = conv_fwd(xa, nra, ca, nullptr, grad, &ona);
      yb = conv_fwd(xb, nrb, cb, nullptr, grad, &onb);
      return;
    }
    ya = tensor(xa.H, xa.W, ca.Cout, grad);
    yb = tensor(xb.H, xb.W, cb.Cout, grad);
    const size_t half = ((size_t)B * chunks * 3 * ca.Cout * 4 + 255) / 256 * 256;
    if (2 * half > need_cpartial) need_cpartial = 2 * half;
    norm_fwd_sizes(ya); norm_fwd_sizes(yb);
    const int Bc = B, dt = dtype, HWo = xa.H * xa.W;
    const bool ha = nra != nullptr, hb = nrb != nullptr;
    const NormL na = ha ? *nra : NormL{}, nb = hb ? *nrb : NormL{};
    const Tn ta = ya, tb = yb;
    Engine* E = this;
    fwd.push_back([=](Ctx& c) {
      float* pa = (float*)(c.arena + E->scr_cpartial);
      float* pb = (float*)(c.arena + E->scr_cpartial + half);
      int rc = pwr_conv_fwd_stats_pair(c.arena + xa.off, c.packs + ca.pack_f, c.params + ca.b, ha ? (float*)(c.arena + na.state) : nullptr,
                                       c.arena + ta.off, pa, c.arena + xb.off, c.packs + cb.pack_f, c.params + cb.b,
                                       hb ? (float*)(c.arena + nb.state) : nullptr, c.arena + tb.off, pb, 1, Bc, xa.H, xa.W, ca.Cin, ca.Cout,
                                       ca.k, dt, c.stream);
      if (rc == PWR_EUNSUPPORTED) {
        rc = pwr_conv_fwd_stats(c.arena + xa.off, c.packs + ca.pack_f, c.params + ca.b, ha ? (float*)(c.arena + na.state) : nullptr, 1, nullptr,
                                c.arena + ta.off, Bc, xa.H, xa.W, ca.Cin, ca.Cout, ca.k, 1, 0, pa, nullptr, nullptr, nullptr, 1, dt, c.stream);
        if (rc) return rc;
        rc = pwr_conv_fwd_stats(c.arena + xb.off, c.packs + cb.pack_f, c.params + cb.b, hb ? (float*)(c.arena + nb.state) : nullptr, 1, nullptr,
                                c.arena + tb.off, Bc, xb.H, xb.W, cb.Cin, cb.Cout, cb.k, 1, 0, pb, nullptr, nullptr, nullptr, 1, dt, c.stream);
      }
      if (rc) return rc;
      // (round 6: ONE finalisation launch for the two heads' norms)
      return pwr_norm_finalize_partial_pair(pa, c.params + ona.gamma, c.params + ona.beta, (float*)(c.arena + ona.state), pb, c.params + onb.gamma,
                                            c.params + onb.beta, (float*)(c.arena + onb.state), chunks, Bc, HWo, ca.Cout, 1e-5f, c.stream);
    });
  }

  // backward of conv_fwd given y.goff complete.  Writes dW (and db if bias_grad), then dgrad into x.goff
  // (accumulating onto x.goff when accumulate_dx).  The caller applies norm_bwd afterwards when nr != null.
  // Returns the slab rows per sample of the norm-backward reductions that the data-gradient launch wrote for `nr`
  // (to be passed to norm_bwd), 0 if it did not.
  int dgrad_stats_chunks(const NormL* nr, const ConvL& cv, const Tn& y, bool accumulate_dx) const {
    if (!nr || accumulate_dx || cv.stride > 2 || !(stats_mask() & 2)) return 0;
    return pwr_conv_stats_chunks(y.H, y.W, cv.Cout, cv.Cin, cv.k, 1, cv.stride == 2 ? 1 : 0, dtype);
  }
  // own_out != null: where the norm backward of `nr` can take the fold (norm_bwd), the slab is a buffer of this layer's own and *own_out is
  // its arena offset (kNoOwn otherwise): pass it on to norm_bwd
  int conv_bwd(const Tn& x, const NormL* nr, const ConvL& cv, const Tn& y, bool bias_grad, bool need_dx, bool accumulate_dx,
               bool handoff = false, size_t* own_out = nullptr) {
    if (own_out) *own_out = kNoOwn;
    const int Bc = B, dt = dtype;
    const bool has_nr = nr != nullptr;
    const NormL n = has_nr ? *nr : NormL{};
    const int M = B * y.H * y.W;
    const int splits = splits_for(M, cv.Cin, cv.Cout, cv.k, cv.stride);
    want_slab(pwr_conv_wgrad_slab_bytes(cv.Cout, cv.Cin, cv.k, splits));
    want_slab((size_t)pwr_colsum_blocks(M) * cv.Cout * 4);
    Engine* E = this;
    // parameter gradients: side stream
    bwd_cur.push_back([=](Ctx& c) {
      return run_on_side(c, [=](Ctx& c2) {
        int rc = pwr_conv_wgrad(c2.arena + x.off, c2.arena + y.goff, has_nr ? (float*)(c2.arena + n.state) : nullptr,
                                1, (float*)(c2.arena + E->scr_slab + c2.slab_off), c2.grads + cv.w, 0,
                                Bc, x.H, x.W, cv.Cin, cv.cin_real, cv.Cout, cv.Cout, cv.k, cv.stride, splits, dt, c2.stream);
        if (rc) return rc;
        if (bias_grad || (E->norm_mode == 1 && !c2.training))
          rc = pwr_colsum_nhwc(c2.arena + y.goff, (float*)(c2.arena + E->scr_slab + c2.slab_off), c2.grads + cv.b, (long long)M, cv.Cout, 0, dt, c2.stream);
        return rc;
      });
    });
    if (!need_dx) return 0;
    // data gradient: main stream
    const int chunks = dgrad_stats_chunks(nr, cv, y, accumulate_dx);
    if (chunks > 0) {
      const size_t need = (size_t)B * chunks * 2 * cv.Cin * 4;
      size_t own = kNoOwn;
      if (handoff) { if (need > need_handoff) need_handoff = need; }
      else if (own_out && fold_ok(cv.Cin)) { own = alloc(need, "nbpart"); *own_out = own; }
      else if (need > need_cpartial) need_cpartial = need;
      const int nm = norm_mode;
      const size_t Engine::*cpart = handoff ? &Engine::scr_handoff : &Engine::scr_cpartial;
      bwd_cur.push_back([=](Ctx& c) {
        if (elim_mask() & 4) return 0;
        const int mode = nm == 0 ? 0 : (c.training ? 1 : 2);
        if (mode == 2)
          return pwr_conv_fwd(c.arena + y.goff, c.packs + cv.pack_d, nullptr, nullptr, 0, nullptr, c.arena + x.goff, nullptr, Bc, y.H, y.W,
                              cv.Cout, cv.Cin, cv.k, 1, cv.stride == 2 ? 1 : 0, dt, c.stream);
        return pwr_conv_fwd_stats(c.arena + y.goff, c.packs + cv.pack_d, nullptr, nullptr, 0, nullptr, c.arena + x.goff, Bc, y.H, y.W,
                                  cv.Cout, cv.Cin, cv.k, 1, cv.stride == 2 ? 1 : 0, nullptr, c.arena + x.off, (float*)(c.arena + n.state),
                                  own != kNoOwn ? (float*)(c.arena + own) : (float*)(c.arena + E->*cpart), 1, dt, c.stream);
      });
      return chunks;
    }
    bwd_cur.push_back([=](Ctx& c) {
      if (elim_mask() & 4) return 0;
      if (cv.stride == 1)
        return pwr_conv_fwd(c.arena + y.goff, c.packs + cv.pack_d, nullptr, nullptr, 0,
                            accumulate_dx ? c.arena + x.goff : nullptr, c.arena + x.goff, nullptr, Bc, y.H, y.W, cv.Cout, cv.Cin, cv.k,
                            1, 0, dt, c.stream);
      return pwr_conv_fwd(c.arena + y.goff, c.packs + cv.pack_d, nullptr, nullptr, 0,
                          accumulate_dx ? c.arena + x.goff : nullptr, c.arena + x.goff, nullptr, Bc, y.H, y.W, cv.Cout, cv.Cin, cv.k, 1,
                          1, dt, c.stream);
    });
    return 0;
  }

  // ---- ResBlock (model.py:6-23)
  struct ResB { NormL na, nb, nc; ConvL ca, cb, cc; Tn t1, t2; };
  // Weight gradients of the fused small-map ResBlocks: while `small_jobs` is set (hourglass(): the region below the 32x32 level),
  // resblock_fused records its three conv layers here instead of pushing a launch + a split-K reduce for each; the region's owner
  // issues them as ONE grouped launch (pwr_conv_wgrad_group) behind the last block's backward kernel.
  struct SmallJob { size_t x_off, dy_off, state_off; long long w; int H, W, Cin, Cout, k; };
  std::vector<SmallJob>* small_jobs = nullptr;
  Op small_group_op(const std::vector<SmallJob>& jobs) {
    const int Bc = B, dt = dtype;
    Engine* E = this;
    std::vector<pwr_wgrad_job> tmp;
    for (auto& j : jobs) tmp.push_back(pwr_wgrad_job{nullptr, nullptr, nullptr, nullptr, j.H, j.W, j.Cin, j.Cin, j.Cout, j.Cout, j.k, 1});
    const size_t bytes = pwr_conv_wgrad_group_slab_bytes(tmp.data(), (int)tmp.size(), B);
    if (!bytes) err = "internal: grouped weight-gradient job list refused";
    want_slab(bytes);
    return [=](Ctx& c) {
      return run_on_side(c, [=](Ctx& c2) {
        std::vector<pwr_wgrad_job> arr;
        for (auto& j : jobs)
          arr.push_back(pwr_wgrad_job{c2.arena + j.x_off, c2.arena + j.dy_off, (const float*)(c2.arena + j.state_off), c2.grads + j.w, j.H, j.W,
                                      j.Cin, j.Cin, j.Cout, j.Cout, j.k, 1});
        return pwr_conv_wgrad_group(arr.data(), (int)arr.size(), (float*)(c2.arena + E->scr_slab + c2.slab_off), Bc, dt, c2.stream);
      });
    };
  }
  Tn resblock(const Tn& x) {
    const bool tr = training;
    ResB r;
    const int Fh = x.C / 2;
    r.na = norm_params(x.C);
    r.ca = conv_params(x.C, Fh, 1, 1, true, true);
    r.nb = norm_params(Fh);
    r.cb = conv_params(Fh, Fh, 3, 1, true, true);
    r.nc = norm_params(Fh);
    r.cc = conv_params(Fh, x.C, 1, 1, true, true);
    if (pwr_resblock_small_supported(x.H, x.W, x.C, norm_mode, dtype)) return resblock_fused(x, r);
    if (lazy_x.valid && lazy_x.x_off == x.off && norm_mode == 0 && src_fuse_on()) {
      norm_fwd_src(x, r.na, lazy_x);
      lazy_x.valid = false;
    } else {
      if (lazy_x.valid) err = "internal: a fused producer is pending in front of an unfused ResBlock (its input would never be written)";
      norm_fwd(x, r.na);
    }
    r.t1 = conv_fwd(x, &r.na, r.ca, nullptr, tr, &r.nb);
    r.t2 = conv_fwd(r.t1, &r.nb, r.cb, nullptr, tr, &r.nc);
    Tn out = conv_fwd(r.t2, &r.nc, r.cc, &x, tr);
    if (tr) {
      // reverse order: pushed first = executed last
      std::vector<Op> blk;
      std::swap(blk, bwd_cur);
      size_t own = kNoOwn;
      int ch = conv_bwd(r.t2, &r.nc, r.cc, out, true, true, false, false, &own);
      norm_bwd(r.t2, r.nc, 0, false, ch, false, 0, own);
      ch = conv_bwd(r.t1, &r.nb, r.cb, r.t2, false, true, false, false, &own);
      norm_bwd(r.t1, r.nb, 0, false, ch, false, 0, own);
      ch = conv_bwd(x, &r.na, r.ca, r.t1, false, true, false, false, &own);
      norm_bwd(x, r.na, out.goff, true, ch, false, 0, own);  // x.g = out.g (skip) + NRbwd(g)
      append_block(blk);
    }
    return out;
  }
  // Small square maps (inner hourglass levels, bf16 + InstanceNorm): the whole block is one launch per direction
  // (csrc/resblock_small.hip).  It writes the same tensors / states as the unfused sequence, so the weight gradients
  // (side stream) are the usual kernels.
  Tn resblock_fused(const Tn& x, ResB r) {
    const bool tr = training;
    const int Bc = B, dt = dtype, Fh = x.C / 2;
    if (tr) { r.t1 = tensor(x.H, x.W, Fh, true); r.t2 = tensor(x.H, x.W, Fh, true); }
    Tn out = tensor(x.H, x.W, x.C, tr);
    const ResB rb = r;
    // the producer of x, if it was left to this launch (hourglass(): the level's max-pool, the inner level's up-sample + add)
    int xmode = 0;
    size_t xa_off = 0, xh_off = 0;
    if (lazy_x.valid && lazy_x.x_off == x.off) { xmode = lazy_x.mode; xa_off = lazy_x.a_off; xh_off = lazy_x.h_off; lazy_x.valid = false; }
    else if (lazy_x.valid) err = "internal: a fused producer (max-pool / up-sample left to a one-launch ResBlock) was not consumed by the block it was registered for";
    fwd.push_back([=](Ctx& c) {
      return pwr_resblock_fwd_small_x(xmode, xmode ? c.arena + xa_off : nullptr, xmode == 2 ? c.arena + xh_off : nullptr,
                                    c.arena + x.off, tr ? c.arena + rb.t1.off : nullptr, tr ? c.arena + rb.t2.off : nullptr, c.arena + out.off,
                                    c.packs + rb.ca.pack_f, c.packs + rb.cb.pack_f, c.packs + rb.cc.pack_f, c.params + rb.ca.b,
                                    c.params + rb.cb.b, c.params + rb.cc.b, c.params + rb.na.gamma, c.params + rb.na.beta,
                                    c.params + rb.nb.gamma, c.params + rb.nb.beta, c.params + rb.nc.gamma, c.params + rb.nc.beta,
                                    (float*)(c.arena + rb.na.state), (float*)(c.arena + rb.nb.state), (float*)(c.arena + rb.nc.state),
                                    Bc, x.H, x.W, x.C, 1e-5f, dt, c.stream);
    });
    if (tr) {
      std::vector<Op> blk;
      std::swap(blk, bwd_cur);
      const size_t bsum = alloc((size_t)B * x.C * 4);           // per-sample column sums of out.g (bias gradient of conv c)
      // neighbours of the block in the hourglass backward left to this launch (hourglass())
      const bool has_up = pend_up.valid;
      const size_t up_goff = pend_up.src_goff;
      pend_up.valid = false;
      PendPool pl{};
      if (pend_pool.valid && pend_pool.x_off == x.off) { pl = pend_pool; pend_pool.valid = false; }
      else if (pend_pool.valid) err = "internal: a fused max-pool backward was not consumed by the block it was registered for";
      if (small_jobs) {
        small_jobs->push_back(SmallJob{r.t2.off, out.goff, r.nc.state, r.cc.w, x.H, x.W, Fh, x.C, 1});
        small_jobs->push_back(SmallJob{r.t1.off, r.t2.goff, r.nb.state, r.cb.w, x.H, x.W, Fh, Fh, 3});
        small_jobs->push_back(SmallJob{x.off, r.t1.goff, r.na.state, r.ca.w, x.H, x.W, x.C, Fh, 1});
      } else if (!has_up) {
        conv_bwd(r.t2, &r.nc, r.cc, out, false, false, false);    // side stream: dW_c (needs only out.g)
      }
      bwd_cur.push_back([=](Ctx& c) {
        if (elim_mask() & 4) return 0;
        return pwr_resblock_bwd_small_x(has_up ? c.arena + up_goff : nullptr, pl.valid ? c.arena + pl.a_off : nullptr,
                                        pl.valid ? c.arena + pl.addend_goff : nullptr, pl.valid ? c.arena + pl.dst_goff : nullptr,
                                        c.arena + out.goff, c.arena + x.off, c.arena + rb.t1.off, c.arena + rb.t2.off, c.arena + x.goff,
                                        c.arena + rb.t1.goff, c.arena + rb.t2.goff, c.packs + rb.cc.pack_d, c.packs + rb.cb.pack_d,
                                        c.packs + rb.ca.pack_d, (float*)(c.arena + rb.na.state), (float*)(c.arena + rb.nb.state),
                                        (float*)(c.arena + rb.nc.state), (float*)(c.arena + rb.na.sums), (float*)(c.arena + rb.nb.sums),
                                        (float*)(c.arena + rb.nc.sums), (float*)(c.arena + bsum), Bc, x.H, x.W, x.C, dt, c.stream);
      });
      if (!small_jobs && has_up) conv_bwd(r.t2, &r.nc, r.cc, out, false, false, false);    // (out.g is written by the launch above)
      if (!small_jobs) {
        conv_bwd(r.t1, &r.nb, r.cb, r.t2, false, false, false);   // side stream: dW_b from t2.g
        conv_bwd(x, &r.na, r.ca, r.t1, false, false, false);      // side stream: dW_a from t1.g
      }
      bwd_cur.push_back([=](Ctx& c) {
        return run_on_side(c, [=](Ctx& c2) {
          int rc = pwr_resblock_param_grads((float*)(c2.arena + rb.na.sums), (float*)(c2.arena + rb.nb.sums), (float*)(c2.arena + rb.nc.sums),
                                            (float*)(c2.arena + bsum), c2.grads + rb.na.gamma, c2.grads + rb.na.beta, c2.grads + rb.nb.gamma,
                                            c2.grads + rb.nb.beta, c2.grads + rb.nc.gamma, c2.grads + rb.nc.beta, c2.grads + rb.cc.b, Bc, x.C,
                                            c2.stream);
          return rc;
        });
      });
      append_block(blk);
    }
    return out;
  }
  // bwd_cur currently holds one block's ops in execution order; `prev` holds blocks built earlier (which must run
  // AFTER this one).  Result: bwd_cur = this block followed by prev.
  void append_block(std::vector<Op>& prev) {
    bwd_cur.insert(bwd_cur.end(), prev.begin(), prev.end());
  }

  // ---- Hourglass (model.py:25-47)
  // (One launch per direction for the whole region below the 32x32 level was built and measured in round 2: bit-identical, not
  // faster -- DESIGN.md section 4 -- and removed in round 3.)
  Tn hourglass(const Tn& x, int lvl) {
    struct Depth { int& d; Depth(int& r) : d(r) { ++d; } ~Depth() { --d; } } depth_guard(hg_depth);
    const bool tr = training;
    scope = "s" + std::to_string(cur_stage) + ".hg" + std::to_string(lvl);
    const int Bc = B, dt = dtype;
    Tn a = resblock(x);
    Tn h0 = tensor(a.H / 2, a.W / 2, a.C, tr);
    // the pooled map's only forward consumer is the next ResBlock: when that is the one-launch kernel of the small maps, it pools on
    // the fly (and writes h0 for the backward pass) instead of a launch of its own
    static const bool fuse_in = PWR_DBG_ENV("PWR_RESBLOCK_FUSE_IN", 1) != 0;
    if (fuse_in && pwr_resblock_small_supported(h0.H, h0.W, h0.C, norm_mode, dtype)) {
      if (lazy_x.valid) { err = "unconsumed fused producer"; }
      lazy_x = LazyX{true, 1, h0.off, a.off, 0};
    } else if (norm_mode == 0 && src_fuse_on() && a.H == 2 * h0.H && a.W == 2 * h0.W) {
      // (round 6) ... and when it is an unfused ResBlock (the 32x32 level), its first norm's statistics launch pools on the fly (norm_fwd_src)
      if (lazy_x.valid) { err = "unconsumed fused producer"; }
      lazy_x = LazyX{true, 1, h0.off, a.off, 0};
    } else {
      fwd.push_back([=](Ctx& c) { return pwr_maxpool_fwd(c.arena + a.off, c.arena + h0.off, Bc, a.H, a.W, a.C, dt, c.stream); });
    }
    // the ops pushed by resblock(x) must run after everything below: take them out, put them back at the end
    std::vector<Op> after_a;
    std::swap(after_a, bwd_cur);
    // everything between this level's pool and its up-sample lives on maps of 16x16 .. 2x2 (fused ResBlocks): their weight
    // gradients go out as grouped launches
    std::vector<SmallJob> jobs;
    const bool own = tr && !small_jobs && PWR_DBG_ENV("PWR_WGRAD_GROUP", 1) != 0 && h0.H <= 16 && pwr_resblock_small_supported(h0.H, h0.W, h0.C, norm_mode, dtype);
    if (own) small_jobs = &jobs;
    Tn out = tensor(a.H, a.W, a.C, tr);
    // backward counterparts of the two fusions: max-pool backward in the store of the block that produces h0's gradient, up-sample
    // backward in the load of the block that consumes h2's (8 launches of 6 - 8 us less on the chain per stage)
    static const bool fuse_bwd = PWR_DBG_ENV("PWR_RESBLOCK_FUSE_BWD", 1) != 0;
    const bool small_inner = tr && fuse_bwd && pwr_resblock_small_supported(h0.H, h0.W, h0.C, norm_mode, dtype);
    if (small_inner) {
      if (pend_pool.valid) err = "unconsumed fused max-pool backward";
      pend_pool = PendPool{true, h0.off, a.off, out.goff, a.goff};
    }
    Tn h1 = lvl > 0 ? hourglass(h0, lvl - 1) : resblock(h0);
    const bool pool_fused = small_inner && !pend_pool.valid;
    if (pend_pool.valid) { err = "internal: fused max-pool backward left pending"; pend_pool.valid = false; }
    std::vector<Op> after_inner;
    std::swap(after_inner, bwd_cur);
    if (small_inner) pend_up = PendUp{true, out.goff};
    Tn h2 = resblock(h1);
    const bool up_fused = small_inner && !pend_up.valid;
    if (pend_up.valid) { err = "internal: fused up-sample backward left pending"; pend_up.valid = false; }
    std::vector<Op> after_h2;
    std::swap(after_h2, bwd_cur);
    if (own) small_jobs = nullptr;
    // `out` of an inner level is consumed by the outer level's output ResBlock only: the same fusion (not for the outermost level --
    // hg_depth == 1 here -- whose result goes to the heads, nor for the big maps)
    if (fuse_in && hg_depth > 1 && a.H >= 4 && pwr_resblock_small_supported(a.H, a.W, a.C, norm_mode, dtype)) {
      if (lazy_x.valid) { err = "unconsumed fused producer"; }
      lazy_x = LazyX{true, 2, out.off, a.off, h2.off};
    } else if (hg_depth > 1 && norm_mode == 0 && src_fuse_on() && a.H == 2 * h2.H && a.W == 2 * h2.W) {
      // (round 6) the outer level's output ResBlock is an unfused one (the 32x32 map): norm_fwd_src adds the up-sampled map while it sums
      if (lazy_x.valid) { err = "unconsumed fused producer"; }
      lazy_x = LazyX{true, 2, out.off, a.off, h2.off};
    } else {
      fwd.push_back([=](Ctx& c) {
        return pwr_upsample_add_fwd(c.arena + h2.off, c.arena + a.off, c.arena + out.off, Bc, h2.H, h2.W, a.H, a.W, a.C, dt, c.stream);
      });
    }
    if (tr) {
      if (!up_fused)
        bwd_cur.push_back([=](Ctx& c) { return pwr_upsample_bwd(c.arena + out.goff, c.arena + h2.goff, Bc, h2.H, h2.W, a.H, a.W, a.C, dt, c.stream); });
      bwd_cur.insert(bwd_cur.end(), after_h2.begin(), after_h2.end());        // resblock h1 -> h2
      bwd_cur.insert(bwd_cur.end(), after_inner.begin(), after_inner.end());  // inner
      for (size_t j0 = 0; own && j0 < jobs.size(); j0 += 24)                  // (<= 24 layers per grouped launch)
        bwd_cur.push_back(small_group_op(std::vector<SmallJob>(jobs.begin() + j0, jobs.begin() + std::min(jobs.size(), j0 + 24))));
      if (!pool_fused)
        bwd_cur.push_back([=](Ctx& c) {
          return pwr_maxpool_bwd(c.arena + a.off, c.arena + h0.goff, c.arena + out.goff, c.arena + a.goff, Bc, a.H, a.W, a.C, dt, c.stream);
        });
      bwd_cur.insert(bwd_cur.end(), after_a.begin(), after_a.end());          // resblock x -> a
    }
    return out;
  }

  // Channel padding of the two narrow NHWC tensors of the network -- the heads' output gradient (J channels) and the stage-input
  // concat (2J + 1 channels).  bf16: up to the 32 / 64 channels of a whole K chunk, so that the convs reading them are served by the
  // LDS-patch kernels (Cin in {32, 64, 128}) instead of the universal implicit-GEMM kernel: at C2 the heads' last data gradient
  // (16 -> 128 channels, 4.8 GFLOP) took 50 - 70 us there, four times per step on the chain.  The weight packs already span whole
  // K chunks (zero-filled), so nothing else changes.  fp32 (K chunks of 16) and wider tensors keep the multiple of 8.
  int pad_narrow(int c) const {
    if (dtype == PWR_BF16 && c <= 64 && PWR_DBG_ENV("PWR_PAD_NARROW", 1)) return c <= 32 ? 32 : 64;
    return (c + 7) / 8 * 8;
  }
  // ---- one regression head (model.py:54-65 / 103-114) ending in an NCHW fp32 map
  struct Head { ConvL c0, c1, c2, c3; NormL n0, n1, n2; Tn h1, h2, h3; size_t gT; };
  // Both heads of a stage (plane: logits z in the arena; depth: the external depth-map output), parameters in named_parameters order
  // (plane head first), the three 128 -> 128 convs of the same depth launched pairwise (conv_fwd_pair)
  void heads_fwd(const Tn& f, size_t z_off, int stage_idx, Head& hp, Head& hd) {
    const bool tr = training;
    const int Bc = B, dt = dtype, Jc = J;
    Head* hs[2] = {&hp, &hd};
    for (int k = 0; k < 2; ++k) {
      Head& h = *hs[k];
      scope = "s" + std::to_string(stage_idx) + (k == 0 ? ".plane" : ".depth");
      h.c0 = conv_params(F, F, ks, 1, true, true, P); h.n0 = norm_params(F);
      h.c1 = conv_params(F, F, ks, 1, true, true, P); h.n1 = norm_params(F);
      h.c2 = conv_params(F, F, ks, 1, true, true, P); h.n2 = norm_params(F);
      h.c3 = conv_params(F, J, ks, 1, true, true);
    }
    scope = "s" + std::to_string(stage_idx) + ".heads";
    conv_fwd_pair(f, nullptr, hp.c0, hp.n0, hp.h1, f, nullptr, hd.c0, hd.n0, hd.h1, tr);
    conv_fwd_pair(hp.h1, &hp.n0, hp.c1, hp.n1, hp.h2, hd.h1, &hd.n0, hd.c1, hd.n1, hd.h2, tr);
    conv_fwd_pair(hp.h2, &hp.n1, hp.c2, hp.n2, hp.h3, hd.h2, &hd.n1, hd.c2, hd.n2, hd.h3, tr);
    {
      // the two last convs (128 -> J, fp32 NCHW maps): ONE launch where the narrow weight-stationary kernel takes the shape, else two
      const Tn ha = hp.h3, hb = hd.h3; const NormL na = hp.n2, nb = hd.n2; const ConvL ca = hp.c3, cb = hd.c3;
      fwd.push_back([=](Ctx& c) {
        float* za = (float*)(c.arena + z_off);
        float* zb = c.out_D[stage_idx];
        int rc = pwr_conv_fwd_nchw_pair(c.arena + ha.off, c.packs + ca.pack_f, c.params + ca.b, (float*)(c.arena + na.state), za,
                                        c.arena + hb.off, c.packs + cb.pack_f, c.params + cb.b, (float*)(c.arena + nb.state), zb,
                                        1, Bc, ha.H, ha.W, ca.Cin, Jc, ca.k, dt, c.stream);
        if (rc != PWR_EUNSUPPORTED) return rc;
        rc = pwr_conv_fwd(c.arena + ha.off, c.packs + ca.pack_f, c.params + ca.b, (float*)(c.arena + na.state),
                          1, nullptr, nullptr, za, Bc, ha.H, ha.W, ca.Cin, Jc, ca.k, 1, 0, dt, c.stream);
        if (rc) return rc;
        return pwr_conv_fwd(c.arena + hb.off, c.packs + cb.pack_f, c.params + cb.b, (float*)(c.arena + nb.state),
                            1, nullptr, nullptr, zb, Bc, hb.H, hb.W, cb.Cin, Jc, cb.k, 1, 0, dt, c.stream);
      });
    }
    const int Jp = pad_narrow(J);
    for (int k = 0; k < 2; ++k) hs[k]->gT = tr ? alloc((size_t)B * P * P * Jp * esz, "gT") : 0;
  }
  // g_nchw_off: arena offset of the fp32 [B,J,N] gradient of the head's output map
  void head_bwd(const Tn& f, const Head& h, size_t g_nchw_off, bool accumulate_df) {
    const int Bc = B, dt = dtype, Jc = J, Pc = P;
    scope = "s" + std::to_string(cur_stage) + (accumulate_df ? ".depth.bwd" : ".plane.bwd");
    const int Jp = pad_narrow(J);
    const int M = B * P * P;
    const int splits = splits_for(M, F, Jp, ks);
    want_slab(pwr_conv_wgrad_slab_bytes(Jp, F, ks, splits));
    Engine* E = this;
    const Tn h3 = h.h3; const NormL n2 = h.n2; const ConvL c3 = h.c3; const size_t gT = h.gT;
    const int nm = norm_mode;
    const int ch3 = (stats_mask() & 2) ? pwr_conv_stats_chunks(P, P, Jp, c3.Cin, c3.k, 1, 0, dtype) : 0;
    const size_t own3 = (ch3 > 0 && fold_ok(c3.Cin)) ? alloc((size_t)B * ch3 * 2 * c3.Cin * 4, "nbpart") : kNoOwn;
    if (ch3 > 0 && own3 == kNoOwn && (size_t)B * ch3 * 2 * c3.Cin * 4 > need_cpartial) need_cpartial = (size_t)B * ch3 * 2 * c3.Cin * 4;
    bwd_cur.push_back([=](Ctx& c) {
      int rc = pwr_nchw_to_nhwc_pad((const float*)(c.arena + g_nchw_off), c.arena + gT, Bc, Jc, Pc * Pc, Jp, dt, c.stream);
      if (rc) return rc;
      rc = pwr_planesum_nchw((const float*)(c.arena + g_nchw_off), (float*)(c.arena + E->scr_S1), c.grads + c3.b, Bc, Jc, Pc * Pc, 0, c.stream);
      if (rc) return rc;
      rc = run_on_side(c, [=](Ctx& c2) {
        return pwr_conv_wgrad(c2.arena + h3.off, c2.arena + gT, (float*)(c2.arena + n2.state), 1,
                              (float*)(c2.arena + E->scr_slab + c2.slab_off), c2.grads + c3.w, 0, Bc, h3.H, h3.W, c3.Cin, c3.Cin, Jp, Jc, c3.k, 1, splits, dt, c2.stream);
      });
      if (rc) return rc;
      const int mode = nm == 0 ? 0 : (c.training ? 1 : 2);
      if (ch3 > 0 && mode != 2)
        return pwr_conv_fwd_stats(c.arena + gT, c.packs + c3.pack_d, nullptr, nullptr, 0, nullptr, c.arena + h3.goff, Bc, Pc, Pc, Jp, c3.Cin,
                                  c3.k, 1, 0, nullptr, c.arena + h3.off, (float*)(c.arena + n2.state),
                                  own3 != kNoOwn ? (float*)(c.arena + own3) : (float*)(c.arena + E->scr_cpartial), 1, dt, c.stream);
      return pwr_conv_fwd(c.arena + gT, c.packs + c3.pack_d, nullptr, nullptr, 0, nullptr, c.arena + h3.goff, nullptr, Bc, Pc, Pc,
                          Jp, c3.Cin, c3.k, 1, 0, dt, c.stream);
    });
    norm_bwd(h.h3, h.n2, 0, false, ch3, false, 0, own3);
    size_t own = kNoOwn;
    int ch = conv_bwd(h.h2, &h.n1, h.c2, h.h3, false, true, false, false, &own);
    norm_bwd(h.h2, h.n1, 0, false, ch, false, 0, own);
    ch = conv_bwd(h.h1, &h.n0, h.c1, h.h2, false, true, false, false, &own);
    norm_bwd(h.h1, h.n0, 0, false, ch, false, 0, own);
    conv_bwd(f, nullptr, h.c0, h.h1, false, true, accumulate_df);
  }

  // Both heads of a stage backwards in LOCK-STEP (round 4): the plane and the depth head run the same three 128 -> 128 convs on different
  // tensors, so their data gradients of one depth go out as ONE launch (pwr_conv_dgrad_stats_pair: a boundary between two full-chip
  // launches of the dominant conv costs 8 - 9 us, tools/launch_bubble.py) and their weight gradients as ONE launch of the
  // wave-specialised kernel + ONE reduce (pwr_conv_wgrad_pair: half the splits per layer -- half the split-K slab traffic -- at twice the
  // K extent).  Same kernels, same per-launch arithmetic as head_bwd() twice; falls back to that when a shape has no pair kernel.
  void heads_bwd(const Tn& f, const Head& hp, const Head& hd, size_t gz_off, size_t gD_off) {
    static const bool on = PWR_DBG_ENV("PWR_HEAD_BWD_PAIR", 1) != 0;
    const int Jp = pad_narrow(J);
    const int ch = (stats_mask() & 2) ? pwr_conv_stats_chunks(P, P, F, F, ks, 1, 0, dtype) : 0;
    const int ch3 = (stats_mask() & 2) ? pwr_conv_stats_chunks(P, P, Jp, F, ks, 1, 0, dtype) : 0;
    if (!on || dtype != PWR_BF16 || norm_mode != 0 || ks != 3 || F != 128 || ch <= 0 || ch3 != ch || (Jp != 32 && Jp != 64) || P % 32) {
      head_bwd(f, hp, gz_off, false);
      head_bwd(f, hd, gD_off, true);
      return;
    }
    scope = "s" + std::to_string(cur_stage) + ".heads.bwd";
    const int Bc = B, dt = dtype, Jc = J, Pc = P, Fc = F, kk = ks;
    const int M = B * P * P;
    const size_t half = ((size_t)B * ch * 2 * F * 4 + 255) / 256 * 256;
    // the norm-backward slabs of the three depths: buffers of their own where the fold applies (norm_bwd), else the shared scratch
    const bool fold = fold_ok(F);
    size_t own[3] = {kNoOwn, kNoOwn, kNoOwn};
    for (int q = 0; q < 3 && fold; ++q) own[q] = alloc(2 * half, "nbpart2");
    const size_t own0 = own[0];
    if (!fold && 2 * half > need_cpartial) need_cpartial = 2 * half;
    const int splits3 = splits_for(M, F, Jp, ks);
    want_slab(pwr_conv_wgrad_slab_bytes(Jp, F, ks, splits3));
    want_slab((size_t)B * J * 4);
    const bool wpair = PWR_DBG_ENV("PWR_HEAD_WGRAD_PAIR", 0) != 0;      // weight gradients of the two heads in one launch (measured: no gain)
    // round 6: the norm backward between two paired data gradients (h3 / n2 and h2 / n1) runs in the staging of the data gradient BELOW it
    // (pwr_conv_dgrad_fold_stats_pair): the apply launch leaves the chain, the dy it would have written goes to a buffer of its own
    // (gdy: a tile's halo reads its neighbours' RAW pixels, so dy cannot replace g in place) which the weight gradients read
    const bool fdg = fold && PWR_DBG_ENV("PWR_HEAD_FOLD_DGRAD", 1) != 0;
    size_t gdy[2][2] = {{0, 0}, {0, 0}};       // [level q][head]: dy of h3 (q = 0) / h2 (q = 1)
    for (int q = 0; q < 2 && fdg; ++q)
      for (int k = 0; k < 2; ++k) gdy[q][k] = alloc((size_t)B * P * P * F * esz, "gdy");
    // round 6: the norm-fed layers' operand materialised once (see below).  Measured: bit-identical and 0.04 ms per step SLOWER (5.196 / 5.235
    // against 5.161 / 5.186 ms, interleaved): the extra 67 MB of traffic per layer beside the chain costs what the loader's arithmetic did.  Off.
    const bool napply = PWR_DBG_ENV("PWR_HEAD_NAPPLY", 0) != 0;
    if (napply && (size_t)B * P * P * F * esz > need_napply) need_napply = (size_t)B * P * P * F * esz;
    const int splits_pair = wpair ? std::max(1, splits_for(M, F, F, ks) / 2) : splits_for(M, F, F, ks);
    want_slab(2 * pwr_conv_wgrad_slab_bytes(F, F, ks, splits_pair));
    Engine* E = this;
    const Head P_ = hp, D_ = hd;
    // The heads' weight gradients are HELD BACK until the heads' data-gradient chain has been issued (Ctx::deferred) and run beside the
    // hourglass backward, whose kernels are small: issued as their operands became ready, two 72-CU weight-gradient launches at a time
    // sat beside the heads' full-chip data gradients and norm backwards (paired data gradient 134 us against 92 alone).  Same launches,
    // same results bit for bit; train step 5.77 -> 5.63 ms (same box, twice; three side streams or more splits on top: slower).
    // (Rounds 1 and 3 measured the opposite with the full-chip weight-gradient kernels of the time: what changed is round 4's few, long
    // workgroups.)  PWR_DEFER_HEADS=0 (debug build): the old order.
    static const int defer_mode = PWR_DBG_ENV("PWR_DEFER_HEADS", 1);
    if (defer_mode) bwd_cur.push_back([=](Ctx& c) { c.defer = true; return 0; });
    // ---- the heads' last convs (F -> J): output gradients to NHWC, bias sums, the two weight gradients, the paired data gradient
    bwd_cur.push_back([=](Ctx& c) {
      const Head* hs[2] = {&P_, &D_};
      const size_t gsrc[2] = {gz_off, gD_off};
      int rc = 0;
      rc = pwr_nchw_to_nhwc_pad_pair((const float*)(c.arena + gsrc[0]), c.arena + P_.gT, (const float*)(c.arena + gsrc[1]), c.arena + D_.gT, Bc, Jc, Pc * Pc,
                                     Jp, dt, c.stream);       // (round 6: one launch for both heads)
      for (int k = 0; k < 2 && !rc; ++k) {
        const Head& h = *hs[k];
        // (the last conv's bias gradient feeds the flat gradient only: side stream, with that stream's slab as scratch -- round 4; it was two
        // launches of 6 us per head on the chain)
        const size_t src = gsrc[k];
        const long long bo = h.c3.b;
        if (!rc) rc = run_on_side(c, [=](Ctx& c2) {
          return pwr_planesum_nchw((const float*)(c2.arena + src), (float*)(c2.arena + E->scr_slab + c2.slab_off), c2.grads + bo, Bc, Jc, Pc * Pc, 0, c2.stream);
        });
      }
      for (int k = 0; k < 2 && !rc; ++k) {
        const Head h = *hs[k];
        rc = run_on_side(c, [=](Ctx& c2) {
          return pwr_conv_wgrad(c2.arena + h.h3.off, c2.arena + h.gT, (float*)(c2.arena + h.n2.state), 1, (float*)(c2.arena + E->scr_slab + c2.slab_off),
                                c2.grads + h.c3.w, 0, Bc, Pc, Pc, Fc, Fc, Jp, Jc, kk, 1, splits3, dt, c2.stream);
        });
      }
      if (rc) return rc;
      float* pa = (float*)(c.arena + (fold ? own0 : E->scr_cpartial));
      float* pb = (float*)(c.arena + (fold ? own0 : E->scr_cpartial) + half);
      rc = pwr_conv_dgrad_stats_pair(c.arena + P_.gT, c.packs + P_.c3.pack_d, c.arena + P_.h3.goff, c.arena + P_.h3.off, (float*)(c.arena + P_.n2.state), pa,
                                     c.arena + D_.gT, c.packs + D_.c3.pack_d, c.arena + D_.h3.goff, c.arena + D_.h3.off, (float*)(c.arena + D_.n2.state), pb,
                                     1, Bc, Pc, Pc, Jp, Fc, kk, dt, c.stream);
      if (rc != PWR_EUNSUPPORTED) return rc;
      rc = pwr_conv_fwd_stats(c.arena + P_.gT, c.packs + P_.c3.pack_d, nullptr, nullptr, 0, nullptr, c.arena + P_.h3.goff, Bc, Pc, Pc, Jp, Fc, kk, 1, 0,
                              nullptr, c.arena + P_.h3.off, (float*)(c.arena + P_.n2.state), pa, 1, dt, c.stream);
      if (rc) return rc;
      return pwr_conv_fwd_stats(c.arena + D_.gT, c.packs + D_.c3.pack_d, nullptr, nullptr, 0, nullptr, c.arena + D_.h3.goff, Bc, Pc, Pc, Jp, Fc, kk, 1, 0,
                                nullptr, c.arena + D_.h3.off, (float*)(c.arena + D_.n2.state), pb, 1, dt, c.stream);
    });
    if (!fdg) norm_bwd_pair(hp.h3, hp.n2, hd.h3, hd.n2, ch, half, own[0]);
    else {       // (its parameter sums still come from the slab: the segment's grouped launch)
      seg_norm_jobs.push_back(NormJob{own[0], false, hp.n2.gamma, hp.n2.beta, P * P, F, ch});
      seg_norm_jobs.push_back(NormJob{own[0] + half, false, hd.n2.gamma, hd.n2.beta, P * P, F, ch});
    }
    // ---- conv depth 2, 1: (x, its norm, conv, y) per head
    struct Lvl { Tn xp, xd, yp, yd; NormL np, nd; ConvL cp, cd; };
    const Lvl lv[2] = {{hp.h2, hd.h2, hp.h3, hd.h3, hp.n1, hd.n1, hp.c2, hd.c2}, {hp.h1, hd.h1, hp.h2, hd.h2, hp.n0, hd.n0, hp.c1, hd.c1}};
    for (int q = 0; q < 2; ++q) {
      const Lvl L = lv[q];
      const size_t ownq = own[q + 1];
      // y's gradient as the weight gradient reads it (dy), and -- folded form -- y's norm (the one ABOVE this conv) with its slab
      const size_t dyp = fdg ? gdy[q][0] : L.yp.goff, dyd = fdg ? gdy[q][1] : L.yd.goff;
      const NormL nyp = q == 0 ? hp.n2 : hp.n1, nyd = q == 0 ? hd.n2 : hd.n1;
      const size_t own_above = own[q];
      bwd_cur.push_back([=](Ctx& c) {
        auto wgrads = [=](Ctx& c) { return run_on_side(c, [=](Ctx& c2) {
          float* slab = (float*)(c2.arena + E->scr_slab + c2.slab_off);
          int r2 = !wpair ? PWR_EUNSUPPORTED
                          : pwr_conv_wgrad_pair(c2.arena + L.xp.off, c2.arena + dyp, (float*)(c2.arena + L.np.state), c2.grads + L.cp.w,
                                                c2.arena + L.xd.off, c2.arena + dyd, (float*)(c2.arena + L.nd.state), c2.grads + L.cd.w, 1, slab, Bc, Pc, Pc,
                                                Fc, Fc, splits_pair, dt, c2.stream);
          if (r2 != PWR_EUNSUPPORTED) return r2;
          if (napply) {
            // the operand relu(norm(x)) written out ONCE on this stream (67 MB of traffic), then the weight gradient's plain form: the same
            // LDS tiles, the same slabs, the same dW -- without the in-LDS norm arithmetic of the loader waves (three kernel rows x 24
            // splits did it over and over: 124.5 us per layer in the step against 84.9 us for the plain form)
            char* na = c2.arena + E->scr_napply + c2.napply_off;
            r2 = pwr_norm_apply(c2.arena + L.xp.off, (float*)(c2.arena + L.np.state), na, 1, Bc, Pc * Pc, Fc, dt, c2.stream);
            if (!r2) r2 = pwr_conv_wgrad(na, c2.arena + dyp, nullptr, 1, slab, c2.grads + L.cp.w, 0, Bc, Pc, Pc, Fc, Fc, Fc, Fc, kk, 1, splits_pair, dt, c2.stream);
            if (!r2) r2 = pwr_norm_apply(c2.arena + L.xd.off, (float*)(c2.arena + L.nd.state), na, 1, Bc, Pc * Pc, Fc, dt, c2.stream);
            if (!r2) r2 = pwr_conv_wgrad(na, c2.arena + dyd, nullptr, 1, slab, c2.grads + L.cd.w, 0, Bc, Pc, Pc, Fc, Fc, Fc, Fc, kk, 1, splits_pair, dt, c2.stream);
            return r2;
          }
          r2 = pwr_conv_wgrad(c2.arena + L.xp.off, c2.arena + dyp, (float*)(c2.arena + L.np.state), 1, slab, c2.grads + L.cp.w, 0, Bc, Pc, Pc, Fc, Fc, Fc,
                              Fc, kk, 1, splits_pair, dt, c2.stream);
          if (r2) return r2;
          return pwr_conv_wgrad(c2.arena + L.xd.off, c2.arena + dyd, (float*)(c2.arena + L.nd.state), 1, slab, c2.grads + L.cd.w, 0, Bc, Pc, Pc, Fc, Fc, Fc,
                                Fc, kk, 1, splits_pair, dt, c2.stream);
        }); };
        int rc = fdg ? 0 : wgrads(c);        // (folded: dy does not exist before the data gradient below has run)
        if (rc) return rc;
        if (elim_mask() & 4) return 0;
        float* pa = (float*)(c.arena + (fold ? ownq : E->scr_cpartial));
        float* pb = (float*)(c.arena + (fold ? ownq : E->scr_cpartial) + half);
        if (fdg) {
          rc = pwr_conv_dgrad_fold_stats_pair(c.arena + L.yp.goff, c.packs + L.cp.pack_d, c.arena + L.xp.goff, c.arena + L.xp.off, (float*)(c.arena + L.np.state), pa,
                                              c.arena + L.yp.off, (float*)(c.arena + nyp.state), (float*)(c.arena + own_above), c.arena + dyp,
                                              c.arena + L.yd.goff, c.packs + L.cd.pack_d, c.arena + L.xd.goff, c.arena + L.xd.off, (float*)(c.arena + L.nd.state), pb,
                                              c.arena + L.yd.off, (float*)(c.arena + nyd.state), (float*)(c.arena + own_above + half), c.arena + dyd,
                                              ch, 1, 1, Bc, Pc, Pc, Fc, Fc, kk, dt, c.stream);
          if (rc) return rc;
          return wgrads(c);
        }
        rc = pwr_conv_dgrad_stats_pair(c.arena + L.yp.goff, c.packs + L.cp.pack_d, c.arena + L.xp.goff, c.arena + L.xp.off, (float*)(c.arena + L.np.state), pa,
                                       c.arena + L.yd.goff, c.packs + L.cd.pack_d, c.arena + L.xd.goff, c.arena + L.xd.off, (float*)(c.arena + L.nd.state), pb,
                                       1, Bc, Pc, Pc, Fc, Fc, kk, dt, c.stream);
        if (rc != PWR_EUNSUPPORTED) return rc;
        rc = pwr_conv_fwd_stats(c.arena + L.yp.goff, c.packs + L.cp.pack_d, nullptr, nullptr, 0, nullptr, c.arena + L.xp.goff, Bc, Pc, Pc, Fc, Fc, kk, 1, 0,
                                nullptr, c.arena + L.xp.off, (float*)(c.arena + L.np.state), pa, 1, dt, c.stream);
        if (rc) return rc;
        return pwr_conv_fwd_stats(c.arena + L.yd.goff, c.packs + L.cd.pack_d, nullptr, nullptr, 0, nullptr, c.arena + L.xd.goff, Bc, Pc, Pc, Fc, Fc, kk, 1, 0,
                                  nullptr, c.arena + L.xd.off, (float*)(c.arena + L.nd.state), pb, 1, dt, c.stream);
      });
      if (!fdg || q == 1) norm_bwd_pair(L.xp, L.np, L.xd, L.nd, ch, half, ownq);      // (h1 / n0: the first convs' data gradients are not the pair kernel)
      else {
        seg_norm_jobs.push_back(NormJob{ownq, false, L.np.gamma, L.np.beta, P * P, F, ch});
        seg_norm_jobs.push_back(NormJob{ownq + half, false, L.nd.gamma, L.nd.beta, P * P, F, ch});
      }
    }
    // (PWR_DEFER_FLUSH=1, debug build: release the held-back side work one op earlier, beside the two first-conv data gradients)
    if (defer_mode && PWR_DBG_ENV("PWR_DEFER_FLUSH", 0) == 1) bwd_cur.push_back(flush_deferred_op());
    // ---- the heads' first convs read the hourglass output f as it is: paired weight gradient (no norm), then f.g = both data gradients
    bwd_cur.push_back([=](Ctx& c) {
      int rc = run_on_side(c, [=](Ctx& c2) {
        float* slab = (float*)(c2.arena + E->scr_slab + c2.slab_off);
        int r2 = !wpair ? PWR_EUNSUPPORTED
                        : pwr_conv_wgrad_pair(c2.arena + f.off, c2.arena + P_.h1.goff, nullptr, c2.grads + P_.c0.w, c2.arena + f.off, c2.arena + D_.h1.goff, nullptr,
                                              c2.grads + D_.c0.w, 1, slab, Bc, Pc, Pc, Fc, Fc, splits_pair, dt, c2.stream);
        if (r2 != PWR_EUNSUPPORTED) return r2;
        r2 = pwr_conv_wgrad(c2.arena + f.off, c2.arena + P_.h1.goff, nullptr, 1, slab, c2.grads + P_.c0.w, 0, Bc, Pc, Pc, Fc, Fc, Fc, Fc, kk, 1, splits_pair, dt, c2.stream);
        if (r2) return r2;
        return pwr_conv_wgrad(c2.arena + f.off, c2.arena + D_.h1.goff, nullptr, 1, slab, c2.grads + D_.c0.w, 0, Bc, Pc, Pc, Fc, Fc, Fc, Fc, kk, 1, splits_pair, dt, c2.stream);
      });
      if (rc) return rc;
      if (elim_mask() & 4) return 0;
      rc = pwr_conv_fwd(c.arena + P_.h1.goff, c.packs + P_.c0.pack_d, nullptr, nullptr, 0, nullptr, c.arena + f.goff, nullptr, Bc, Pc, Pc, Fc, Fc, kk, 1, 0, dt, c.stream);
      if (rc) return rc;
      return pwr_conv_fwd(c.arena + D_.h1.goff, c.packs + D_.c0.pack_d, nullptr, nullptr, 0, c.arena + f.goff, c.arena + f.goff, nullptr, Bc, Pc, Pc, Fc, Fc, kk, 1, 0, dt,
                          c.stream);
    });
    if (defer_mode) bwd_cur.push_back(flush_deferred_op());
  }
  static Op flush_deferred_op() {
    return [](Ctx& c) {
      c.defer = false;
      std::vector<Ctx::Deferred> d;
      d.swap(c.deferred);
      int rc = 0;
      // ONE fork for the whole batch: every held-back op depends on the chain as it stands here, so both side streams wait for one event
      // and the ops go out without an event pair each (a dozen ops: ~25 API calls less in a stretch where the host has just issued the
      // heads' chain and the hourglass backward's small kernels are next)
      const bool batch = PWR_DBG_ENV("PWR_FLUSH_BATCH", 1) != 0 && c.use_side && c.n_side > 0 && !(elim_mask() & 9) && !d.empty();
      if (batch) {
        hipEvent_t ev = c.ev_fork[c.fork_rr];
        c.fork_rr = (c.fork_rr + 1) % c.n_fork;
        hipEventRecord(ev, (hipStream_t)c.stream);
        for (int k = 0; k < c.n_side; ++k) hipStreamWaitEvent(c.side[k], ev, 0);
      }
      for (size_t i = 0; i < d.size() && !rc; ++i) rc = run_on_side(c, d[i].fn, batch, d[i].light);
      return rc;
    };
  }

  // ---------------------------------------------------------------- whole network
  bool build() {
    const bool tr = training;
    const int Bc = B, dt = dtype, Jc = J, Pc = P, Fc = F, S = 2 * P, N = P * P, meth = method;
    if (F % 32 || F < 32) { err = "features must be a multiple of 32 for the MI355X engine"; return false; }
    if (stages > 8) { err = "at most 8 stages"; return false; }
    if (J > 47) { err = "at most 47 joints"; return false; }
    Engine* E = this;
    scope = "stem";
    // ---- stem (model.py:164-187)
    std::vector<ConvL> sc;
    std::vector<NormL> sn;
    std::vector<Tn> sy;
    {
      ConvL c0 = conv_params(1, 32, ks, 1, false, false);
      NormL n0 = norm_params(32);
      Tn y0 = tensor(S, S, 32, tr);
      fwd.push_back([=](Ctx& c) { return pwr_stem_conv_fwd(c.img, c.params + c0.w, c.params + c0.b, c.arena + y0.off, Bc, S, 32, E->ks, dt, c.stream); });
      norm_fwd(y0, n0);
      fwd_first_pack_op = fwd.size();            // everything from here on may read a weight pack
      sc.push_back(c0); sn.push_back(n0); sy.push_back(y0);
      int cch = 32;
      while (cch < F) {
        const int nx = 2 * cch < F ? 2 * cch : F;
        ConvL cv = conv_params(cch, nx, ks, 1, true, true);
        NormL nn = norm_params(nx);
        Tn y = conv_fwd(sy.back(), &sn.back(), cv, nullptr, tr, &nn);
        sc.push_back(cv); sn.push_back(nn); sy.push_back(y);
        cch = nx;
      }
      ConvL cl = conv_params(F, F, ks, 2, true, true);
      NormL nl = norm_params(F);
      Tn yl = conv_fwd(sy.back(), &sn.back(), cl, nullptr, tr, &nl);
      sc.push_back(cl); sn.push_back(nl); sy.push_back(yl);
    }
    std::vector<Op> stem_bwd;
    int stem_in_chunks = 0;   // slab rows per sample that stage 0's input-conv data gradient writes for the stem's last norm
    // ---- stages
    struct StageRec { size_t z, gz, gDt, gH, gD, gwp; long long w_off; };
    std::vector<StageRec> recs(stages);
    std::vector<std::vector<Op>> stage_bwd(stages);
    const Tn ystem = sy.back();
    const NormL nstem = sn.back();
    for (int s = 0; s < stages; ++s) {
      StageRec& R = recs[s];
      cur_stage = s;
      scope = "s" + std::to_string(s) + ".in";
      Tn x0;
      ConvL cin;
      const int Cp = pad_narrow(2 * J + 1);
      Tn xc;  // NHWC concat [B,P,P,Cp] for stages >= 1
      if (s == 0) {
        cin = conv_params(F, F, 1, 1, true, true);
      } else {
        cin = conv_params(2 * J + 1, F, 1, 1, true, true);   // packs are built from the 2J+1 real channels ...
        cin.Cin = Cp;                                        // ... the tensor it reads has Cp (zero-padded) channels
      }
      // hourglass params come before the heads' in named_parameters order; plane_regression.w comes first in its module
      if (s == 0) {
        x0 = conv_fwd(ystem, &nstem, cin, nullptr, tr);
      } else {
        xc = tensor(P, P, Cp, tr);
        const int sp = s - 1;
        const Tn xcc = xc;
        fwd.push_back([=](Ctx& c) { return pwr_cat_to_nhwc(c.out_p[sp], c.out_D[sp], c.label, c.arena + xcc.off, Bc, Jc, N, Cp, dt, c.stream); });
        x0 = conv_fwd(xc, nullptr, cin, nullptr, tr);
      }
      std::vector<Op> saved;
      std::swap(saved, bwd_cur);  // (empty) -- keep bwd_cur clean for the hourglass
      Tn f = hourglass(x0, level);
      std::vector<Op> hg_bwd;
      std::swap(hg_bwd, bwd_cur);
      R.w_off = method == 0 ? take_param(J) : -1;
      scope = "s" + std::to_string(s) + ".dec";
      R.z = alloc((size_t)B * J * N * 4, "z");
      Head hp, hd;
      heads_fwd(f, R.z, s, hp, hd);
      const size_t zoff = R.z;
      const long long woff = R.w_off;
      fwd.push_back([=](Ctx& c) {
        return pwr_decode_fwd((const float*)(c.arena + zoff), c.out_D[s], c.label, c.mask, woff >= 0 ? c.params + woff : nullptr, c.out_p[s],
                              c.out_uvd[s], Bc, Jc, Pc, meth, c.stream);
      });
      if (tr) {
        scope = "s" + std::to_string(s) + ".dec";
        R.gz = alloc((size_t)B * J * N * 4, "gz"); R.gDt = alloc((size_t)B * J * N * 4, "gDt");
        R.gH = alloc((size_t)B * J * N * 4, "gH"); R.gD = alloc((size_t)B * J * N * 4, "gD");
        R.gwp = alloc((size_t)B * J * 4, "gwp");
        const StageRec Rc = R;
        const bool last = s == stages - 1;
        const size_t zero_uvd = alloc((size_t)B * J * 3 * 4);
        // decoder backward: coupling gradients (from stage s+1, already in gH/gD) + external gradients
        bwd_cur.push_back([=](Ctx& c) {
          const float* gH = nullptr; const float* gD = nullptr;
          const long long n = (long long)Bc * Jc * N;
          int rc = 0;
          if (!last) {
            gH = (const float*)(c.arena + Rc.gH); gD = (const float*)(c.arena + Rc.gD);
            if (c.g_p[s]) rc = pwr_add_inplace(c.g_p[s], c.arena + Rc.gH, n, PWR_F32, c.stream);
            if (!rc && c.g_D[s]) rc = pwr_add_inplace(c.g_D[s], c.arena + Rc.gD, n, PWR_F32, c.stream);
            if (rc) return rc;
          } else {
            gH = c.g_p[s]; gD = c.g_D[s];
          }
          const float* gU = c.g_uvd[s];
          if (!gU) {
            hipMemsetAsync(c.arena + zero_uvd, 0, (size_t)Bc * Jc * 3 * 4, (hipStream_t)c.stream);
            gU = (const float*)(c.arena + zero_uvd);
          }
          rc = pwr_decode_bwd(c.out_p[s], (const float*)(c.arena + Rc.z), c.out_D[s], c.label, c.mask, woff >= 0 ? c.params + woff : nullptr,
                              c.out_uvd[s], gH, gD, gU, (float*)(c.arena + Rc.gz), (float*)(c.arena + Rc.gDt),
                              woff >= 0 ? (float*)(c.arena + Rc.gwp) : nullptr, Bc, Jc, Pc, meth, c.stream);
          if (rc || woff < 0) return rc;
          // (the soft-argmax temperature's gradient: a one-block reduce that only feeds the flat gradient -- side stream)
          return run_on_side(c, [=](Ctx& c2) { return pwr_decode_gw_reduce((const float*)(c2.arena + Rc.gwp), c2.grads + woff, Bc, Jc, 0, c2.stream); });
        });
        heads_bwd(f, hp, hd, R.gz, R.gDt);
        bwd_cur.insert(bwd_cur.end(), hg_bwd.begin(), hg_bwd.end());
        // stage-input conv backward
        if (s == 0) {
          stem_in_chunks = conv_bwd(ystem, &nstem, cin, x0, true, true, false, /*handoff=*/true);
        } else {
          const int sp = s - 1;
          const StageRec Rp = recs[sp];
          const Tn xcc = xc;
          conv_bwd(xc, nullptr, cin, x0, true, true, false);   // dW [F][2J+1], db, and d(concat) into xc.goff
          bwd_cur.push_back([=](Ctx& c) {
            return pwr_nhwc_to_cat_grad(c.arena + xcc.goff, (float*)(c.arena + Rp.gH), (float*)(c.arena + Rp.gD), Bc, Jc, N, Cp, dt, c.stream);
          });
        }
        flush_norm_jobs();
        std::swap(stage_bwd[s], bwd_cur);
      }
    }
    scope = "stem.bwd";
    if (tr) {
      if (!bwd_cur.empty()) { err = "internal: backward list not empty before the stem"; return false; }
      const int ns = (int)sc.size();
      want_slab((size_t)pwr_stem_conv_wgrad_blocks(B, S) * 32 * ks * ks * 4);
      // the gradient of the stem output comes from stage 0's 1x1 input conv (built above, run in the previous segment):
      // stem_in_chunks is what that launch really writes (its conv_bwd's return value), in the hand-off buffer
      int ch = stem_in_chunks;
      size_t own = kNoOwn;
      for (int i = ns - 1; i >= 1; --i) {
        norm_bwd(sy[i], sn[i], 0, false, ch, /*handoff=*/i == ns - 1, 0, own);
        ch = conv_bwd(sy[i - 1], &sn[i - 1], sc[i], sy[i], false, true, false, false, &own);
      }
      norm_bwd(sy[0], sn[0], 0, false, ch, false, 0, own);
      const Tn y0 = sy[0]; const ConvL c0 = sc[0];
      bwd_cur.push_back([=](Ctx& c) {
        return run_on_side(c, [=](Ctx& c2) {
          return pwr_stem_conv_wgrad(c2.img, c2.arena + y0.goff, (float*)(c2.arena + E->scr_slab + c2.slab_off), c2.grads + c0.w, 0, Bc, S, 32, E->ks, dt, c2.stream);
        });
      });
      flush_norm_jobs();
      std::swap(stem_bwd, bwd_cur);
    }
    if (pcur != poff.size()) { err = "parameter table longer than the network"; return false; }
    if (lazy_x.valid) { err = "internal: a fused producer (max-pool / up-sample) is still pending at the end of the plan"; return false; }
    if (!err.empty()) return false;
    // shared scratch
    scope = "scratch";
    scr_partial = alloc(need_partial, "partial");
    scr_S1 = alloc(need_sc, "S1"); scr_S2 = alloc(need_sc, "S2");
    scr_cpartial = alloc(need_cpartial, "cpartial");
    scr_handoff = alloc(need_handoff, "stem_handoff");
    need_napply = (need_napply + 255) / 256 * 256;
    scr_napply = alloc(need_napply * Ctx::kMaxSide, "napply");
    ctx.napply_stride = need_napply;
    scr_slab_bytes = (scr_slab_bytes + 255) / 256 * 256;
    scr_slab = alloc(scr_slab_bytes * Ctx::kMaxSide, "slab");
    ctx.slab_stride = scr_slab_bytes;
    // segments in execution order: last stage first, stem last
    if (tr) {
      for (int s = stages - 1; s >= 0; --s) bwd.push_back(stage_bwd[s]);
      bwd.push_back(stem_bwd);
    }
    // descriptor table lives in the pack buffer
    desc_dev_off = alloc_pack(descs.size() * sizeof(PackDescHost));
    return true;
  }
};

thread_local std::string g_last_error;

}  // namespace

extern "C" const char* pwr_last_error(void) { return g_last_error.c_str(); }

// cfg: {joints, stage, label_size, features, level, kernel_size, norm_mode (0 instance, 1 batch), heatmap_method}
extern "C" void* pwr_engine_create(const int* cfg, int B, int dtype, int training, const long long* param_off,
                                   const long long* param_numel, int n_params, const long long* buffer_off, int n_buffers) {
  Engine* e = new Engine();
  e->J = cfg[0]; e->stages = cfg[1]; e->P = cfg[2]; e->F = cfg[3]; e->level = cfg[4]; e->ks = cfg[5]; e->norm_mode = cfg[6];
  e->method = cfg[7];
  e->B = B; e->dtype = dtype; e->training = training; e->esz = dtype == PWR_BF16 ? 2 : 4;
  e->poff.assign(param_off, param_off + n_params);
  e->pnum.assign(param_numel, param_numel + n_params);
  if (buffer_off) e->boff.assign(buffer_off, buffer_off + n_buffers);
  if (!e->build()) {
    g_last_error = e->err.empty() ? "engine build failed" : e->err;
    delete e;
    return nullptr;
  }
  return e;
}

extern "C" void pwr_engine_destroy(void* h) {
  Engine* e = (Engine*)h;
  if (e->ev_pack) hipEventDestroy(e->ev_pack);
  for (int k = 0; k < e->ctx.n_side; ++k) hipStreamSynchronize(e->ctx.side[k]);    // (shared streams: drained, not destroyed)
  delete e;
}
#ifdef PWR_DEBUG_BUILD
// Debugging aid: the arena layout as text lines "offset bytes tag".  Returns the number of bytes needed (incl. the NUL);
// writes at most cap bytes.
extern "C" size_t pwr_engine_layout(void* h, char* buf, size_t cap) {
  Engine* e = (Engine*)h;
  std::string s;
  for (auto& r : e->layout) s += std::to_string(r.off) + " " + std::to_string(r.bytes) + " " + r.tag + "\n";
  if (buf && cap) { const size_t n = s.size() + 1 < cap ? s.size() + 1 : cap; memcpy(buf, s.c_str(), n); buf[n - 1] = 0; }
  return s.size() + 1;
}
extern "C" void pwr_engine_set_join(void* h, int each_segment) { ((Engine*)h)->join_each_segment = each_segment != 0; }
#endif   // PWR_DEBUG_BUILD
#ifdef PWR_DEBUG_BUILD
// Debugging aid: per-scope chain time.  set_timing(1) starts collecting (events on the caller's stream at every change of scope in the op
// lists); timing_report synchronises the device, writes lines "phase<TAB>scope<TAB>total ms<TAB>intervals" and clears the collection.
extern "C" void pwr_engine_set_timing(void* h, int on) { ((Engine*)h)->timing = on != 0; }
extern "C" size_t pwr_engine_timing_report(void* h, char* buf, size_t cap) {
  Engine* e = (Engine*)h;
  hipDeviceSynchronize();
  std::vector<std::pair<std::string, std::pair<double, int>>> acc;
  for (size_t i = 0; i + 1 < e->timed.size(); ++i) {
    if (!e->timed[i].tag || e->timed[i].phase != e->timed[i + 1].phase) continue;       // (a closing mark, or the next call's first mark)
    float ms = 0.f;
    hipEventElapsedTime(&ms, e->timed[i].ev, e->timed[i + 1].ev);
    const std::string key = e->timed[i].phase + "\t" + e->timed[i].tag;
    size_t k = 0;
    while (k < acc.size() && acc[k].first != key) ++k;
    if (k == acc.size()) acc.push_back({key, {0.0, 0}});
    acc[k].second.first += ms; acc[k].second.second += 1;
  }
  for (auto& t : e->timed) hipEventDestroy(t.ev);
  e->timed.clear();
  std::string s;
  for (auto& a : acc) s += a.first + "\t" + std::to_string(a.second.first) + "\t" + std::to_string(a.second.second) + "\n";
  if (buf && cap) { const size_t n = s.size() + 1 < cap ? s.size() + 1 : cap; memcpy(buf, s.c_str(), n); buf[n - 1] = 0; }
  return s.size() + 1;
}
#endif

extern "C" size_t pwr_engine_arena_bytes(void* h) { return ((Engine*)h)->arena_bytes; }
extern "C" size_t pwr_engine_pack_bytes(void* h) { return ((Engine*)h)->pack_bytes; }
extern "C" int pwr_engine_num_segments(void* h) { return (int)((Engine*)h)->bwd.size(); }
extern "C" int pwr_engine_num_launch_ops(void* h, int which) {
  Engine* e = (Engine*)h;
  if (which == 0) return (int)e->fwd.size();
  int n = 0;
  for (auto& s : e->bwd) n += (int)s.size();
  return n;
}
extern "C" size_t pwr_engine_desc_offset(void* h) { return ((Engine*)h)->desc_dev_off; }
extern "C" size_t pwr_engine_desc_bytes(void* h) { return ((Engine*)h)->descs.size() * sizeof(PackDescHost); }
extern "C" int pwr_engine_get_descs(void* h, void* host_dst) {
  Engine* e = (Engine*)h;
  memcpy(host_dst, e->descs.data(), e->descs.size() * sizeof(PackDescHost));
  return 0;
}

extern "C" int pwr_engine_bind(void* h, void* arena, void* packs, const float* params, float* grads, float* buffers) {
  Engine* e = (Engine*)h;
  e->ctx.arena = (char*)arena; e->ctx.packs = (char*)packs; e->ctx.params = params; e->ctx.grads = grads; e->ctx.buffers = buffers;
  return 0;
}

// Re-pack every conv weight from the flat fp32 parameters (one launch).  The descriptor table must have been
// uploaded to packs + pwr_engine_desc_offset().
extern "C" int pwr_engine_pack(void* h, void* stream) {
  Engine* e = (Engine*)h;
  return pwr_pack_weights(e->ctx.params, e->ctx.packs, e->ctx.packs + e->desc_dev_off, (int)e->descs.size(), stream);
}
// The same re-pack as PART OF the next pwr_engine_forward: issued on a side stream behind everything already on the forward's stream, beside
// the stem's first conv (which reads the fp32 parameters) and its norm statistics; the forward waits for it in front of the first launch
// that reads a pack.  ~28 us less on the critical path of every forward (the pack of all 88 layers is one launch of that length).
extern "C" int pwr_engine_pack_beside_forward(void* h) {
  ((Engine*)h)->pack_pending = true;
  return 0;
}

// outs: host array of 3*stage device pointers {heatmaps_s, depthmaps_s, uvd_s}
extern "C" int pwr_engine_forward(void* h, const float* img, const float* label, const float* mask, void* const* outs, int training,
                                  void* stream) {
  Engine* e = (Engine*)h;
  Ctx& c = e->ctx;
  c.img = img; c.label = label; c.mask = mask; c.stream = stream; c.training = training;
  for (int s = 0; s < e->stages; ++s) {
    c.out_p[s] = (float*)outs[3 * s]; c.out_D[s] = (float*)outs[3 * s + 1]; c.out_uvd[s] = (float*)outs[3 * s + 2];
  }
  e->generation++;
  auto run = [&](void* st) -> int {
    c.stream = st;
    bool pack_wait = false;
    if (e->pack_pending) {
      e->pack_pending = false;
      if (!c.attached) side_pool_attach(c, (hipStream_t)st);
      // (measured, round 6: beside = 1 makes INFERENCE 4 % slower -- 1.70 against 1.63 ms, two interleaved runs -- and leaves the train step
      // where it was: the two cross-stream dependencies per forward cost more than the 28-us launch they hide.  Off in the product.)
      static const bool beside = PWR_DBG_ENV("PWR_PACK_BESIDE", 0) != 0;
      if (beside && c.use_side && c.n_side > 0 && c.n_fork > 0 && e->fwd_first_pack_op < e->fwd.size()) {
        if (!e->ev_pack && hipEventCreateWithFlags(&e->ev_pack, hipEventDisableTiming) != hipSuccess) e->ev_pack = nullptr;
      }
      if (beside && e->ev_pack && c.use_side && c.n_side > 0 && c.n_fork > 0 && e->fwd_first_pack_op < e->fwd.size()) {
        // the pack must follow everything already on the caller's stream (the optimizer update, the previous step's readers of the packs)
        hipEvent_t ev = c.ev_fork[c.fork_rr];
        c.fork_rr = (c.fork_rr + 1) % c.n_fork;
        hipEventRecord(ev, (hipStream_t)st);
        hipStreamWaitEvent(c.side[0], ev, 0);
        int rc = pwr_pack_weights(c.params, c.packs, c.packs + e->desc_dev_off, (int)e->descs.size(), c.side[0]);
        if (rc) return rc;
        hipEventRecord(e->ev_pack, c.side[0]);
        pack_wait = true;
      } else {
        int rc = pwr_pack_weights(c.params, c.packs, c.packs + e->desc_dev_off, (int)e->descs.size(), st);
        if (rc) return rc;
      }
    }
    // hand-off counters of the fused norm kernels live at the head of the partial scratch: zero once per call
    // (only the hand-off form of the statistics kernel reads them -- debug build's PWR_NORM_FUSE; the product's forward issued this fill,
    // a 5-us launch at the head of every pass, for nobody until round 6)
    static const bool handoff_counters = PWR_DBG_ENV("PWR_NORM_FUSE", 0) != 0;
    if (handoff_counters && e->need_partial) hipMemsetAsync(c.arena + e->scr_partial, 0, 8192 < e->need_partial ? 8192 : e->need_partial, (hipStream_t)st);
    for (size_t i = 0; i < e->fwd.size(); ++i) {
      if (pack_wait && i == e->fwd_first_pack_op) { hipStreamWaitEvent((hipStream_t)st, e->ev_pack, 0); pack_wait = false; }
#ifdef PWR_DEBUG_BUILD
      if (e->timing && (i == 0 || e->fwd[i].tag != e->fwd[i - 1].tag)) e->mark("forward", e->fwd[i].tag, st);
#endif
      int rc = e->fwd[i](c);
      if (rc) { char b[96]; snprintf(b, sizeof b, "forward op %zu failed with %d", i, rc); g_last_error = b; return rc; }
    }
#ifdef PWR_DEBUG_BUILD
    if (e->timing) e->mark("forward", nullptr, st);
#endif
    return 0;
  };
  // (hipGraph capture + replay of this launch list was measured in round 1: 8.5 vs 7.7 ms per step on ROCm 7.2 -- the host issues
  // a step's launches in 2.6 ms, the step is not launch-bound -- and removed in round 3.)
  const int rc = run(stream);
  c.stream = stream;
  return rc;
}
extern "C" long long pwr_engine_generation(void* h) { return ((Engine*)h)->generation; }

// Make `waiter` wait until everything the backward segments issued so far have written is complete: the data-gradient chain on
// `chain_stream` (the stream the segments were issued on) AND the parameter-gradient kernels on the engine's side streams.  Called by the
// data-parallel mode right after segment k has been issued, for the stream its all-reduce of that segment's gradient slice runs on.
extern "C" int pwr_engine_wait_segment(void* h, void* chain_stream, void* waiter) {
  Engine* e = (Engine*)h;
  Ctx& c = e->ctx;
  if (!c.attached) return 0;            // (no backward has run yet)
  hipError_t er = hipSuccess;
  for (int k = 0; c.use_side && k < c.n_side && er == hipSuccess; ++k) {
    er = hipEventRecord(c.ev_join[k], c.side[k]);
    if (er == hipSuccess) er = hipStreamWaitEvent((hipStream_t)waiter, c.ev_join[k], 0);
  }
  if (er == hipSuccess && chain_stream != waiter) {
    hipEvent_t ev = c.n_fork ? c.ev_fork[c.fork_rr] : nullptr;
    if (ev) {
      c.fork_rr = (c.fork_rr + 1) % c.n_fork;
      er = hipEventRecord(ev, (hipStream_t)chain_stream);
      if (er == hipSuccess) er = hipStreamWaitEvent((hipStream_t)waiter, ev, 0);
    } else {
      er = hipStreamSynchronize((hipStream_t)chain_stream);      // (no side streams, no event pool: rare debug configuration)
    }
  }
  return (int)er;
}

// gouts: host array of 3*stage device pointers (NULL = zero gradient).  Runs backward segment `seg`
// (0 = last stage ... num_segments-1 = stem); segment 0 also zeroes the flat gradient buffer first.
extern "C" int pwr_engine_backward(void* h, const void* const* gouts, int seg, long long n_grad_floats, void* stream) {
  Engine* e = (Engine*)h;
  Ctx& c = e->ctx;
  if (!e->training || seg < 0 || seg >= (int)e->bwd.size()) return PWR_EINVAL;
  c.stream = stream;
  for (int s = 0; s < e->stages; ++s) {
    c.g_p[s] = (const float*)gouts[3 * s]; c.g_D[s] = (const float*)gouts[3 * s + 1]; c.g_uvd[s] = (const float*)gouts[3 * s + 2];
  }
  if (!c.attached) side_pool_attach(c, (hipStream_t)stream);
  auto run = [&](void* st) -> int {
    c.stream = st;
    c.side_rr = 0;
    c.defer = false; c.deferred.clear();
    if (seg == 0) {
      hipError_t er = hipMemsetAsync(c.grads, 0, (size_t)n_grad_floats * 4, (hipStream_t)st);
      if (er != hipSuccess) return (int)er;
    }
    auto& ops = e->bwd[seg];
    int rc = 0;
    for (size_t i = 0; i < ops.size() && !rc; ++i) {
#ifdef PWR_DEBUG_BUILD
      if (e->timing && (i == 0 || ops[i].tag != ops[i - 1].tag)) e->mark("backward segment " + std::to_string(seg), ops[i].tag, st);
#endif
      rc = ops[i](c);
      if (rc) { char b[96]; snprintf(b, sizeof b, "backward seg %d op %zu failed with %d", seg, i, rc); g_last_error = b; }
    }
    // join: the caller's stream waits for the side streams after the LAST segment only (the optimizer, and the next forward, need every
    // parameter gradient).  Whoever needs an earlier segment's gradients complete -- the data-parallel all-reduce of that segment's
    // slice -- makes ITS stream wait with pwr_engine_wait_segment(); the chain does not stop for it (round 4: a join per segment cost
    // 0.06 - 0.14 ms of a 5.8 ms step; it was kept in rounds 2 - 3 beside the fix of round 1's non-reproducible step -- the packed-f32
    // form, DESIGN.md section 2 -- and measured then at 1 %).
#ifdef PWR_DEBUG_BUILD
    if (e->timing) e->mark("backward segment " + std::to_string(seg), nullptr, st);
#endif
    // INVARIANTS this rests on: (1) no chain op of segment k + 1 writes a buffer that a side-stream op of segment k still reads -- arena
    // offsets are handed out once per plan by a bump allocator (alloc(): no gradient or activation buffer is ever aliased across segments)
    // and each side stream has its own split-K slab; (2) every consumer of a segment's parameter gradients before the last segment calls
    // pwr_engine_wait_segment.  A segment that FAILED joins at once: the next forward must not start beside side work still in flight.
    const bool join_now = e->join_each_segment || seg + 1 == (int)e->bwd.size() || rc != 0;
    for (int k = 0; join_now && c.use_side && k < c.n_side; ++k) {
      hipEventRecord(c.ev_join[k], c.side[k]);
      hipStreamWaitEvent((hipStream_t)st, c.ev_join[k], 0);
    }
    return rc;
  };
  const int rc = run(stream);
  c.stream = stream;
  return rc;
}
