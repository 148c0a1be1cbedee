"""Split-count sweep of the 3x3 weight gradient at the shapes of the big-map ResBlocks and the stem (kernel + reduce, us).
    python tools/wgrad_splits.py"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pixelwiseregression_amd import kernels as K
dev = "cuda:0"

def timeit(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

for (b, h, cin, cout, normed) in ((32, 64, 64, 64, True), (32, 32, 64, 64, True), (32, 128, 32, 64, True), (32, 128, 64, 128, True), (32, 64, 128, 128, True), (32, 64, 128, 128, False)):
    x = torch.randn(b, h, h, cin, device=dev).to(torch.bfloat16); dy = torch.randn(b, h, h, cout, device=dev).to(torch.bfloat16)
    st = K.norm_stats(x, torch.ones(cin, device=dev), torch.zeros(cin, device=dev), mode=0) if normed else None
    res = {}
    for s in (40, 56, 80, 120, 160, 240, 320):
        if b * h * h // 32 // s < 8: continue
        res[s] = round(timeit(lambda: K.conv_wgrad(x, dy, cout, 3, 1, norm=st, relu_in=True, splits=s)), 1)
    print(json.dumps({"shape": "%dx%dx%dx%d->%d%s" % (b, h, h, cin, cout, " norm" if normed else ""), "us_by_splits": res}))
