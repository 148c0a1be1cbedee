"""Phase timeline of the 3x3 patch conv (heads shape): per-workgroup s_memtime stamps -> where the time goes, per CU."""
import os, sys, torch, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dbglib  # noqa: F401  (the debug build: pwr_debug.h entry points, PWR_* experiment switches)
from pixelwiseregression_amd import kernels as K, _lib
dev = "cuda:0"
B, P, F_ = 32, 64, 128
x = torch.randn(B, P, P, F_, device=dev).to(torch.bfloat16)
w = torch.randn(F_, F_, 3, 3, device=dev) * 0.03
pack = K.pack_conv(w, 0, K.BF16)
st = K.norm_stats(x, torch.ones(F_, device=dev), torch.zeros(F_, device=dev), mode=0)
bias = torch.zeros(F_, device=dev)
nwg = B * (P // 4) * (P // 32)
stamps = torch.zeros(nwg, 8, dtype=torch.int64, device=dev)
l = _lib.lib()
for _ in range(3): K.conv_fwd(x, pack, F_, 3, 1, bias=bias, norm=st)
torch.cuda.synchronize()
l.pwr_debug_set_stamps(stamps.data_ptr())
K.conv_fwd(x, pack, F_, 3, 1, bias=bias, norm=st)
torch.cuda.synchronize()
l.pwr_debug_set_stamps(None)
s = stamps.cpu()
t0 = int(s[:, 0].min())
rel = (s[:, :5] - t0).double()
print("workgroups", nwg, " kernel span (s_memtime ticks)", int(rel[:, 4].max()))
names = ["start", "patch loaded+written", "after staging barrier", "K loop done", "end"]
for i in range(1, 5):
    d = rel[:, i] - rel[:, i - 1]
    print("  %-24s mean %8.0f  min %8.0f  max %8.0f" % (names[i], d.mean(), d.min(), d.max()))
print("  start times: first 512 mean %.0f max %.0f; rest mean %.0f min %.0f max %.0f" % (rel[:512, 0].mean(), rel[:512, 0].max(), rel[512:, 0].mean(), rel[512:, 0].min(), rel[512:, 0].max()))
hw = s[:, 6]; xcc = s[:, 7] & 0xF
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
key = (xcc * 1000 + se * 100 + sh * 10 + cu)
per = collections.Counter(key.tolist())
print("  distinct (xcc,se,sh,cu):", len(per), " workgroups per CU: min %d max %d" % (min(per.values()), max(per.values())))
# timeline of one CU
k0 = key[0].item()
idx = (key == k0).nonzero().flatten().tolist()
for i in idx: print("   wg %4d on CU %d:" % (i, k0), [int(v) for v in rel[i].tolist()])
