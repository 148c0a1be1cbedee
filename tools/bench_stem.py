"""Timing of the stem's stride-2 conv (forward, universal kernel) and its data gradient (mode 1) at the C2 shape."""
import sys, os, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixelwiseregression_amd import kernels as K

dev = "cuda:0"
B, C = 32, 128
x = torch.randn(B, 128, 128, C, device=dev).to(torch.bfloat16)
dy = torch.randn(B, 64, 64, C, device=dev).to(torch.bfloat16)
w = torch.randn(C, C, 3, 3, device=dev) * 0.03
pf = K.pack_conv(w, 0, K.BF16)
pd = K.pack_conv(w, 2, K.BF16)

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

flops = 2.0 * B * 64 * 64 * C * C * 9
t = timeit(lambda: K.conv_fwd(x, pf, C, 3, 2))
print(json.dumps({"kernel": "stride-2 conv fwd", "us": t, "TFLOPs": flops / t / 1e6}))
t = timeit(lambda: K.conv_fwd(dy, pd, C, 3, 1, mode=1))
print(json.dumps({"kernel": "stride-2 conv dgrad (mode 1)", "us": t, "TFLOPs": flops / t / 1e6}))
