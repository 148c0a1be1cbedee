"""Timing by elimination inside the nine-tap wave-specialised weight gradient (debug build, PWR_WGRAD9W_DBG; results are WRONG by construction):
1 the MFMA waves only keep the barriers, 2 no DMA after the prologue, 3 both (the loop skeleton: barriers, waits, the loaders' pass).
C2 heads shape, kernel + reduce (the reduce is ~12 us at 60 splits, ~5 at 18)."""
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
for splits in (60, 18):
    res = {}
    for rnd in range(2):
        for norm, nm in ((st, "norm"), (None, "no norm")):
            for dbg in (0, 1, 2, 3):
                os.environ["PWR_WGRAD9W_DBG"] = str(dbg)
                res.setdefault("%s dbg=%d" % (nm, dbg), []).append(round(timeit(lambda: K.conv_wgrad(x, dy, 128, 3, 1, norm=norm, splits=splits)), 1))
    os.environ["PWR_WGRAD9W_DBG"] = "0"
    print(json.dumps({"splits": splits, **res}), flush=True)
