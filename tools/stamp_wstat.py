"""Phase timeline of conv3x3_wstat_kernel (heads shape): per-workgroup s_memtime stamps (debug build) -> where a persistent workgroup's time goes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dbglib  # noqa: F401
from pixelwiseregression_amd import kernels as K, _lib
dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
STATS = len(sys.argv) > 2 and sys.argv[2] == "stats"
NB = len(sys.argv) > 2 and sys.argv[2] == "nb"
NAR = len(sys.argv) > 2 and sys.argv[2] == "nar"     # the narrow form (128 -> 14, fp32 NCHW): "half A" = the tile's K loop, "half B" = its stores
P, F_ = 64, 128
x = torch.randn(B, P, P, F_, device=dev).to(torch.bfloat16)
w = torch.randn(14 if NAR else F_, F_, 3, 3, device=dev) * 0.03
pack = K.pack_conv(w, 1 if NB else 0, K.BF16, frag="std" not in sys.argv and not NAR)
nby = torch.randn(B, P, P, F_, device=dev).to(torch.bfloat16)
st = K.norm_stats(x, torch.ones(F_, device=dev), torch.zeros(F_, device=dev), mode=0)
bias = torch.zeros(F_, device=dev)
nwg = 256
stamps = torch.zeros(nwg, 32, dtype=torch.int64, device=dev)
l = _lib.lib()
def conv():
    if NAR:
        return K.conv_fwd(x, pack, 14, 3, 1, bias=bias[:14].contiguous(), norm=st, nhwc_out=False, nchw_out=True)
    if NB:
        return K.conv_fwd_stats(x, pack, F_, 3, 1, nb_y=nby, nb_state=st)
    if STATS:
        return K.conv_fwd_stats(x, pack, F_, 3, 1, bias=bias, norm=st)
    return K.conv_fwd(x, pack, F_, 3, 1, bias=bias, norm=st)
for _ in range(3): conv()
torch.cuda.synchronize()
l.pwr_debug_set_stamps(stamps.data_ptr())
conv()
torch.cuda.synchronize()
l.pwr_debug_set_stamps(None)
s = stamps.cpu()
t0 = int(s[:, 0].min())
ntile = B * 32 // nwg
print("workgroups", nwg, "tiles per workgroup", ntile, " kernel span (s_memtime ticks = shader cycles?)", int((s[:, 31] - t0).max()))
print("  start skew: mean %.0f max %.0f" % ((s[:, 0] - t0).double().mean(), (s[:, 0] - t0).max()))
print("  prologue (weights + first patch + barrier): mean %.0f min %.0f max %.0f" % tuple(f((s[:, 1] - s[:, 0]).double()) for f in (torch.mean, torch.min, torch.max)))
print("  prologue parts: start -> patch loads issued %.0f -> weights issued + patch written %.0f -> barrier %.0f -> weights landed %.0f" % tuple(float(v.double().mean()) for v in (s[:, 25] - s[:, 0], s[:, 26] - s[:, 25], s[:, 27] - s[:, 26], s[:, 1] - s[:, 27])))
prev = s[:, 1]
for k in range(min(ntile, 9)):
    a, b_, c = s[:, 2 + 3 * k], s[:, 3 + 3 * k], s[:, 4 + 3 * k]
    print("  tile %d: half A %.0f  half B %.0f  barrier %.0f   (min/max of A: %.0f / %.0f)" % (k, (a - prev).double().mean(), (b_ - a).double().mean(), (c - b_).double().mean(), (a - prev).min(), (a - prev).max()))
    prev = c
print("  tail epilogue: mean %.0f" % (s[:, 31] - prev).double().mean())
clk = ((s[:, 31] - s[:, 0]).double() / (s[:, 30] - s[:, 29]).double() * 100.0)
print("  workgroup life: mean %.0f cycles = %.2f us at the %.0f MHz (min %.0f max %.0f) shader clock it measured" % ((s[:, 31] - s[:, 0]).double().mean(), ((s[:, 30] - s[:, 29]).double().mean() / 100.0), clk.mean(), clk.min(), clk.max()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): conv()
e1.record(); torch.cuda.synchronize()
print("  us per launch (incl. host allocation):", e0.elapsed_time(e1) / 20 * 1e3)
