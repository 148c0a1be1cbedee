"""What a launch boundary costs between two full-chip conv launches: 2 x (B = 32) against 1 x (B = 64) of the dominant conv (same
workgroups, same rounds).    python tools/launch_bubble.py"""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixelwiseregression_amd import kernels as K
dev = "cuda:0"
w = torch.randn(128, 128, 3, 3, device=dev) * 0.03
pack = K.pack_conv(w, 0, K.BF16)
bias = torch.zeros(128, device=dev)
def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
x64 = torch.randn(64, 64, 64, 128, device=dev).to(torch.bfloat16)
xa, xb = x64[:32].contiguous(), x64[32:].contiguous()
st64 = K.norm_stats(x64, torch.ones(128, device=dev), torch.zeros(128, device=dev))
sta, stb = st64[:, :32].contiguous(), st64[:, 32:].contiguous()
for rep in range(3):
    t2 = timeit(lambda: (K.conv_fwd(xa, pack, 128, 3, 1, bias=bias, norm=sta), K.conv_fwd(xb, pack, 128, 3, 1, bias=bias, norm=stb)))
    t1 = timeit(lambda: K.conv_fwd(x64, pack, 128, 3, 1, bias=bias, norm=st64))
    print(json.dumps({"two launches of B=32 (us)": round(t2, 2), "one launch of B=64 (us)": round(t1, 2), "per boundary (us)": round(t2 - t1, 2)}))
