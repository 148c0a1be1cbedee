import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixelwiseregression_amd import kernels as K, _lib
dev = "cuda:0"; l = _lib.lib()
B, P, F_ = 32, 64, 128
x = torch.randn(B, P, P, F_, device=dev).to(torch.bfloat16)
w = torch.randn(F_, F_, 3, 3, device=dev) * 0.03
pack = K.pack_conv(w, 0, K.BF16)
st = K.norm_stats(x, torch.ones(F_, device=dev), torch.zeros(F_, device=dev), mode=0)
bias = torch.zeros(F_, device=dev)
l.pwr_debug_set_pingpong(1)
for _ in range(3): K.conv_fwd(x, pack, F_, 3, 1, bias=bias, norm=st)
stamps = torch.zeros(256 * 16, 8, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
l.pwr_debug_set_stamps(stamps.data_ptr())
K.conv_fwd(x, pack, F_, 3, 1, bias=bias, norm=st)
torch.cuda.synchronize()
l.pwr_debug_set_stamps(None)
s = stamps.cpu().view(256, 16, 8)
for wg in (0, 100, 255):
    print("wg", wg)
    t0 = int(s[wg, 0, 0])
    for h in range(5):
        row = s[wg, h, :7]
        if int(row[0]) == 0: continue
        print("   h %d: start %8d  steps 0-6-12-18-24-30-35: %s" % (h, int(row[0]) - t0, [int(row[k + 1] - row[k]) for k in range(6)]))
