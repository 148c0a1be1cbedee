"""Timing by elimination inside the wave-specialised weight gradient (debug build, PWR_WGRAD3W_DBG; results are WRONG by construction):
1 no norm arithmetic, 2 no stores of the tile pass, 4 no tile pass at all.  C2 heads shape, 80 splits, kernel + reduce."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dbglib  # noqa: F401
import torch
from pixelwiseregression_amd import kernels as K
dev = "cuda:0"
x = torch.randn(32, 64, 64, 128, device=dev).to(torch.bfloat16)
dy = torch.randn(32, 64, 64, 128, device=dev).to(torch.bfloat16)
st = K.norm_stats(x, torch.ones(128, device=dev), torch.zeros(128, device=dev), mode=0)
def timeit(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
res = {}
for rnd in range(2):
    for dbg in (0, 1, 2, 3, 4):
        os.environ["PWR_WGRAD3W_DBG"] = str(dbg)
        res.setdefault("norm dbg=%d" % dbg, []).append(round(timeit(lambda: K.conv_wgrad(x, dy, 128, 3, 1, norm=st, splits=80)), 1))
    os.environ["PWR_WGRAD3W_DBG"] = "0"
    res.setdefault("no norm", []).append(round(timeit(lambda: K.conv_wgrad(x, dy, 128, 3, 1, norm=None, splits=80)), 1))
print(json.dumps(res))
