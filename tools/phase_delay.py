"""Phase experiment on the dominant conv (3x3 128 -> 128 @64x64, B = 32, norm + ReLU prologue): the two workgroups of a CU start together and
stay in phase; hold the one in the odd wave slots back by D cycles (debug build, pwr_debug_set_delay) and time the launch.
   python tools/phase_delay.py [B = 32]"""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dbglib  # noqa: F401
from pixelwiseregression_amd import kernels as K, _lib
dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
P, F_ = 64, 128
x = torch.randn(B, P, P, F_, device=dev).to(torch.bfloat16)
w = torch.randn(F_, F_, 3, 3, device=dev) * 0.03
pack = K.pack_conv(w, 0, K.BF16)
st = K.norm_stats(x, torch.ones(F_, device=dev), torch.zeros(F_, device=dev), mode=0)
bias = torch.zeros(F_, device=dev)
l = _lib.lib()
fl = 2.0 * B * P * P * F_ * F_ * 9


def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for rnd in range(2):
    for d in (0, 1024, 2048, 4096, 6144, 8192, 12288, 16384):
        l.pwr_debug_set_delay(d)
        t = timeit(lambda: K.conv_fwd(x, pack, F_, 3, 1, bias=bias, norm=st))
        t2 = timeit(lambda: K.conv_fwd(x, pack, F_, 3, 1))
        print(json.dumps({"delay_cycles": d, "fwd_form_us": round(t, 2), "dgrad_form_us": round(t2, 2), "frac_fwd": round(fl / t / 1e6 / 2500, 4)}), flush=True)
l.pwr_debug_set_delay(0)
