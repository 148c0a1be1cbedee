"""Elimination inside the LDS-DMA weight gradient (debug build: PWR_WGRAD3D_DBG bits 1 no MFMA, 2 no fragment reads, 4 no DMA after the
prologue, 8 no slab stores) at the C2 heads shape, kernel + reduce, 80 splits.   python tools/wgrad_probe.py"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dbglib  # noqa: F401
import torch
from pixelwiseregression_amd import kernels as K

dev = "cuda:0"
B, P, F_ = 32, 64, 128
x = torch.randn(B, P, P, F_, device=dev).to(torch.bfloat16)
dy = torch.randn(B, P, P, F_, device=dev).to(torch.bfloat16)

def timeit(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

for rep in range(2):
    for dbg, name in ((0, "product"), (1, "no MFMA"), (3, "no fragment reads, no MFMA"), (4, "no DMA after the prologue"), (8, "no slab stores"),
                      (12, "no DMA, no slab stores"), (15, "skeleton: barriers and waits only")):
        os.environ["PWR_WGRAD3D_DBG"] = str(dbg)
        t = timeit(lambda: K.conv_wgrad(x, dy, F_, 3, 1, norm=None, splits=80))
        print(json.dumps({"PWR_WGRAD3D_DBG": dbg, "variant": name, "us_kernel_plus_reduce": round(t, 2)}))
