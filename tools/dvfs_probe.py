"""Is the dominant conv limited by the chip's power management?  The same launch on random operands and on all-zero operands (identical
instruction stream and memory traffic; MI355X_MICROARCH.md 'DVFS give-back': zero operands draw less power and hold a higher clock)."""
import sys, os, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixelwiseregression_amd import kernels as K
dev = "cuda:0"
B, P, F_ = 32, 64, 128
def timeit(fn, iters=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for name, gen in (("random", lambda *s: torch.randn(*s, device=dev)), ("zeros", lambda *s: torch.zeros(*s, device=dev)), ("random again", lambda *s: torch.randn(*s, device=dev))):
    x = gen(B, P, P, F_).to(torch.bfloat16)
    dy = gen(B, P, P, F_).to(torch.bfloat16)
    pack = K.pack_conv(gen(F_, F_, 3, 3) * 0.03, 0, K.BF16)
    print(json.dumps({"operands": name, "conv3x3 128->128 (no prologue) us": round(timeit(lambda: K.conv_fwd(x, pack, F_, 3, 1)), 2),
                      "wgrad3 LDS-DMA + reduce us": round(timeit(lambda: K.conv_wgrad(x, dy, F_, 3, 1, norm=None, splits=80)), 2)}))
