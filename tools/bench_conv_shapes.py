"""The dominant conv (3x3, 128 -> 128, norm + ReLU prologue, bf16; csrc/conv_wstat.hip) at the heads' shapes of BASELINE C2, C3 and C5: what the
prologue of a persistent workgroup (weights + first patch) costs at 4 / 8 / 64 tiles per workgroup.   python tools/bench_conv_shapes.py"""
import sys, os, json, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pixelwiseregression_amd import kernels as K
dev = "cuda:0"
for (B, P) in ((32, 64), (64, 64), (128, 128)):
    x = torch.randn(B, P, P, 128, device=dev).to(torch.bfloat16)
    w = torch.randn(128, 128, 3, 3, device=dev) * 0.03
    pack = K.pack_conv(w, 0, K.BF16, frag=True)
    st = K.norm_stats(x, torch.ones(128, device=dev), torch.zeros(128, device=dev), mode=0)
    bias = torch.zeros(128, device=dev)
    y = torch.empty_like(x)
    l = K._lib.lib()
    def f():
        K._lib.check(l.pwr_conv_fwd(K._p(x), K._p(pack), K._p(bias), K._p(st), 1, None, K._p(y), None, B, P, P, 128, 128, 3, 1, 0, K._dt(x), K._s(x)), "conv")
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e-3
    fl = 2.0 * B * P * P * 128 * 128 * 9
    print(json.dumps({"shape": "B=%d %dx%d 128->128 (+norm prologue)" % (B, P, P), "us": t * 1e6, "TFLOPs": fl / t / 1e12, "frac_of_2500": fl / t / 2.5e15}))
