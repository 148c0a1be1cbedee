"""Timing by elimination of conv_wgrad3_kernel<2,2,1,2> at the C2 head shape (needs PWR_LIB=libpwr_hip_w3dbg.so and PWR_WGRAD3_DBG)."""
import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixelwiseregression_amd import kernels as K
dev = "cuda:0"
B, P, F_ = 32, 64, 128
x = torch.randn(B, P, P, F_, device=dev).to(torch.bfloat16)
dy = torch.randn(B, P, P, F_, device=dev).to(torch.bfloat16)
st = K.norm_stats(x, torch.ones(F_, device=dev), torch.zeros(F_, device=dev), mode=0)
flops = 2.0 * B * P * P * F_ * F_ * 9
def timeit(fn, iters=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
print(json.dumps({"dbg": os.environ.get("PWR_WGRAD3_DBG", "0"), "us_with_norm": timeit(lambda: K.conv_wgrad(x, dy, F_, 3, 1, norm=st, splits=80)),
                  "us_no_norm": timeit(lambda: K.conv_wgrad(x, dy, F_, 3, 1, splits=80))}))
