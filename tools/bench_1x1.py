"""Isolated timing of the universal conv kernel on the chain's 64x64 shapes (1x1 convs, stride-2 stem conv, small-K data gradients)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixelwiseregression_amd import kernels as K
dev = "cuda:0"
B = 32
def timeit(fn, iters=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
def case(name, H, cin, cout, k, stride=1, residual=False, norm=True, mode=0, stats=False):
    x = torch.randn(B, H, H, cin, device=dev).to(torch.bfloat16)
    w = torch.randn(cout, cin, k, k, device=dev) * 0.05
    pk = K.pack_conv(w, 0, K.BF16)
    st = K.norm_stats(x, torch.ones(cin, device=dev), torch.zeros(cin, device=dev)) if norm else None
    Ho = H // stride
    res = torch.randn(B, Ho, Ho, cout, device=dev).to(torch.bfloat16) if residual else None
    bias = torch.zeros(cout, device=dev)
    if stats: fn = lambda: K.conv_fwd_stats(x, pk, cout, k, stride, bias=bias, norm=st, residual=res)
    else: fn = lambda: K.conv_fwd(x, pk, cout, k, stride, bias=bias, norm=st, residual=res)
    t = timeit(fn)
    by = x.numel() * 2 + B * Ho * Ho * cout * 2 * (2 if residual else 1)
    fl = 2.0 * B * Ho * Ho * cin * cout * k * k
    print("%-44s %7.1f us   %6.2f TB/s   %6.1f TFLOP/s" % (name, t, by / t / 1e6, fl / t / 1e6))
case("1x1 128->64 @64 (+NR, stats)", 64, 128, 64, 1, stats=True)
case("1x1 128->64 @64 (+NR)", 64, 128, 64, 1)
case("1x1 64->128 @64 (+NR, +residual)", 64, 64, 128, 1, residual=True)
case("1x1 64->128 @64 (plain: dgrad form)", 64, 64, 128, 1, norm=False)
case("1x1 128->128 @64 (+NR)", 64, 128, 128, 1)
case("3x3 s2 128->128 @128->64 (+NR, stats)", 128, 128, 128, 3, stride=2, stats=True)
case("1x1 128->64 @32", 32, 128, 64, 1, stats=True)
case("1x1 64->128 @32 (+residual)", 32, 64, 128, 1, residual=True)
case("3x3 128->64 @128 (plain: the stem's data gradient)", 128, 128, 64, 3, norm=False)
case("3x3 64->32 @128 (plain)", 128, 64, 32, 3, norm=False)
case("3x3 64->128 @128 (+NR, stats)", 128, 64, 128, 3, stats=True)
def case_tr2(name):
    dy = torch.randn(B, 64, 64, 128, device=dev).to(torch.bfloat16)
    w = torch.randn(128, 128, 3, 3, device=dev) * 0.03
    pk = K.pack_conv(w, 2, K.BF16)
    t = timeit(lambda: K.conv_fwd(dy, pk, 128, 3, 1, mode=1))
    print("%-44s %7.1f us   %6.1f TFLOP/s" % (name, t, 2.0 * B * 64 * 64 * 128 * 128 * 9 / t / 1e6))
case_tr2("stride-2 data gradient 128<-128 @64->128")
