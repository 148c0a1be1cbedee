"""K steps of 64 pixels in the three-tap weight gradient (debug build, PWR_WGRAD3_KP64 bits): correctness against F.conv2d's float64 weight
gradient on the normalised bf16 operand, and isolated timing at the C2 shapes.   PWR_WGRAD3_KP64=15 python tools/kp64_check.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dbglib  # noqa: F401
import torch, torch.nn.functional as F
from pixelwiseregression_amd import kernels as K
dev = "cuda:0"


def q(t): return t.to(torch.bfloat16).double()


def check(B, H, W, Cin, Cout, cr, splits):
    g = torch.Generator().manual_seed(H * 1000 + Cin)
    x = torch.randn(B, Cin, H, W, generator=g, dtype=torch.float64); dy = torch.randn(B, Cout, H, W, generator=g, dtype=torch.float64)
    dy[:, cr:] = 0
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev).to(torch.bfloat16); dyd = dy.permute(0, 2, 3, 1).contiguous().to(dev).to(torch.bfloat16)
    gamma, beta = (1 + 0.3 * torch.randn(Cin, generator=g)).float().to(dev), (0.3 * torch.randn(Cin, generator=g)).float().to(dev)
    st = K.norm_stats(xd, gamma, beta, mode=0)
    dw = K.conv_wgrad(xd, dyd, cr, 3, 1, norm=st, relu_in=True, splits=splits)
    dw2 = K.conv_wgrad(xd, dyd, cr, 3, 1, norm=st, relu_in=True, splits=splits)
    mean, scale, shift = (st[i].double().cpu()[:, :, None, None] for i in (0, 2, 3))
    xin = q(torch.relu((q(x) - mean) * scale + shift))
    w = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(xin, w, None, padding=1).backward(q(dy))
    ref = w.grad[:cr]
    err = float((dw.double().cpu() - ref).abs().max() / ref.abs().max())
    return {"shape": (B, H, W, Cin, Cout, cr, splits), "rel_err": err, "repeatable": bool(torch.equal(dw, dw2))}


def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


print("PWR_WGRAD3_KP64 =", os.environ.get("PWR_WGRAD3_KP64", "0"))
for case in ((2, 64, 64, 128, 16, 14, 7), (3, 8, 128, 32, 64, 64, 5), (2, 64, 64, 64, 64, 64, 9), (2, 16, 64, 64, 128, 128, 3), (5, 64, 64, 128, 32, 21, 80)):
    print(json.dumps(check(*case)))
B = 32
for (tag, H, Cin, Cout, cr, sp) in (("heads' last conv 128->16", 64, 128, 16, 14, 80), ("stem 32->64 @128", 128, 32, 64, 64, 168), ("stem 64->128 @128", 128, 64, 128, 128, 80),
                                    ("3x3 64->64 @64", 64, 64, 64, 64, 80), ("3x3 64->64 @32", 32, 64, 64, 64, 80)):
    x = torch.randn(B, H, H, Cin, device=dev).to(torch.bfloat16); dy = torch.randn(B, H, H, Cout, device=dev).to(torch.bfloat16)
    st = K.norm_stats(x, torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev), mode=0)
    ts = [timeit(lambda: K.conv_wgrad(x, dy, cr, 3, 1, norm=st, splits=sp)) for _ in range(3)]
    print(json.dumps({"layer": tag, "splits": sp, "us": [round(t, 1) for t in ts]}))
