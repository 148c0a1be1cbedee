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

for rep in range(0 if (len(sys.argv) > 1 and sys.argv[1] == 'child') else 2):
    for dbg, name in ((0, "product"), (1, "no MFMA"), (3, "no fragment reads, no MFMA"), (4, "no DMA after the prologue"), (8, "no slab stores"),
                      (12, "no DMA, no slab stores"), (15, "skeleton: barriers and waits only")):
        os.environ["PWR_WGRAD3D_DBG"] = str(dbg)
        t = timeit(lambda: K.conv_wgrad(x, dy, F_, 3, 1, norm=None, splits=80))
        print(json.dumps({"PWR_WGRAD3D_DBG": dbg, "variant": name, "us_kernel_plus_reduce": round(t, 2)}))

# ---- the in-LDS norm variant against the register-staged kernel: bit-identical results, and the time of both (kernel + reduce).
# PWR_WGRAD3_DMA is read once per process: the two variants run in child processes
if len(sys.argv) > 1 and sys.argv[1] == "child":
    res = {}
    for (b, h, w_, cin, cout, splits) in ((32, 64, 64, 128, 128, 80), (4, 64, 64, 128, 128, 80), (3, 20, 96, 64, 64, 24), (2, 32, 32, 128, 64, 8), (2, 8, 32, 64, 128, 3),
                                          (5, 64, 64, 128, 128, 37), (8, 128, 128, 64, 128, 40), (8, 128, 128, 64, 64, 40)):
        g = torch.Generator(device="cpu").manual_seed(b * 1000 + cin)
        xx = torch.randn(b, h, w_, cin, generator=g).to(dev).to(torch.bfloat16)
        dd = torch.randn(b, h, w_, cout, generator=g).to(dev).to(torch.bfloat16)
        gamma = (1 + 0.3 * torch.randn(cin, generator=g)).to(dev); beta = (0.3 * torch.randn(cin, generator=g)).to(dev)
        st = K.norm_stats(xx, gamma, beta, mode=0)
        out = K.conv_wgrad(xx, dd, cout, 3, 1, norm=st, relu_in=True, splits=splits)
        t = timeit(lambda: K.conv_wgrad(xx, dd, cout, 3, 1, norm=st, relu_in=True, splits=splits))
        res["%dx%dx%dx%d->%d s%d" % (b, h, w_, cin, cout, splits)] = round(t, 2)
        torch.save(out.cpu(), "/tmp/wg_%s_%d_%d_%d_%d_%d.pt" % (os.environ.get("PWR_WGRAD3_DMA"), b, h, w_, cin, cout))
    print(json.dumps(res))
else:
    import subprocess, glob
    # elimination inside the in-LDS norm variant (main shape)
    xx = torch.randn(32, 64, 64, 128, device=dev).to(torch.bfloat16); dd = torch.randn(32, 64, 64, 128, device=dev).to(torch.bfloat16)
    st_ = K.norm_stats(xx, torch.ones(128, device=dev), torch.zeros(128, device=dev), mode=0)
    os.environ["PWR_WGRAD3_DMA"] = "2"
    for rep in range(2):
        for dbg, name in ((0, "norm variant, product"), (64, "raw stores instead of the norm arithmetic"), (128, "no norm arithmetic, no stores"), (1, "no MFMA"), (129, "no MFMA, no norm arithmetic / stores"), (4, "no DMA after the prologue"), (8, "no slab stores")):
            os.environ["PWR_WGRAD3D_DBG"] = str(dbg)
            t = timeit(lambda: K.conv_wgrad(xx, dd, 128, 3, 1, norm=st_, relu_in=True, splits=80))
            print(json.dumps({"PWR_WGRAD3D_DBG": dbg, "variant": name, "us_kernel_plus_reduce": round(t, 2)}))
    os.environ["PWR_WGRAD3D_DBG"] = "0"
    for v in ("1", "2"):
        env = dict(os.environ, PWR_WGRAD3_DMA=v, PWR_WGRAD3D_DBG="0")
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True)
        print("PWR_WGRAD3_DMA=" + v, r.stdout.splitlines()[-1] if r.stdout.strip() else r.stderr[-2000:])
    for f1 in sorted(glob.glob("/tmp/wg_1_*.pt")):
        a, b_ = torch.load(f1), torch.load(f1.replace("/tmp/wg_1_", "/tmp/wg_2_"))
        print(os.path.basename(f1), "bit-identical" if torch.equal(a, b_) else "DIFFERENT max %g of %g" % (float((a - b_).abs().max()), float(a.abs().max())))
