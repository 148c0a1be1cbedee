"""Latency study: one small 3x3 conv (64->64 at 2x2 .. 16x16, B=32) on the universal kernel."""
import sys, os, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixelwiseregression_amd import kernels as K
dev = "cuda:0"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
for (H, C, k) in ((2, 64, 3), (4, 64, 3), (8, 64, 3), (16, 64, 3), (2, 128, 1), (16, 128, 1), (64, 128, 1)):
    B = 32
    x = torch.randn(B, H, H, C, device=dev).to(torch.bfloat16)
    co = 64
    w = torch.randn(co, C, k, k, device=dev) * 0.05
    pack = K.pack_conv(w, 0, K.BF16)
    st = K.norm_stats(x, torch.ones(C, device=dev), torch.zeros(C, device=dev), mode=0)
    for nrm in (None, st):
        for _ in range(3): K.conv_fwd(x, pack, co, k, 1, norm=nrm)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): K.conv_fwd(x, pack, co, k, 1, norm=nrm)
        e1.record(); torch.cuda.synchronize()
        print(json.dumps({"H": H, "Cin": C, "k": k, "prologue": nrm is not None, "us": e0.elapsed_time(e1) / iters * 1e3}))
