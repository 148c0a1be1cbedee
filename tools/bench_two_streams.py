"""Does the 3x3 patch conv overlap with itself?  N launches on one stream vs N/2 + N/2 on two streams (plane / depth head)."""
import os, sys, torch, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixelwiseregression_amd import kernels as K
dev = "cuda:0"
B, P, F_ = 32, 64, 128
xs = [torch.randn(B, P, P, F_, device=dev).to(torch.bfloat16) for _ in range(2)]
w = torch.randn(F_, F_, 3, 3, device=dev) * 0.03
pack = K.pack_conv(w, 0, K.BF16)
sts = [K.norm_stats(x, torch.ones(F_, device=dev), torch.zeros(F_, device=dev), mode=0) for x in xs]
bias = torch.zeros(F_, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
N = 200
def run(two):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(N):
        k = i & 1
        with torch.cuda.stream(s2 if (two and k) else s1):
            K.conv_fwd(xs[k], pack, F_, 3, 1, bias=bias, norm=sts[k])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e6
for _ in range(2):
    print("one stream %.1f us/launch   two streams %.1f us/launch" % (run(False), run(True)))
