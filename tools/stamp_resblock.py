"""Phase timeline of the one-launch ResBlock kernels of the small maps (csrc/resblock_small*.{hip,inc}): s_memtime stamps of thread 0 at the
phase boundaries (debug build) -> where a workgroup's (= a sample's) time goes, per map size.      python tools/stamp_resblock.py [B = 32]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dbglib  # noqa: F401
from pixelwiseregression_amd import kernels as K, _lib
dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
C = 128
l = _lib.lib()
torch.manual_seed(0)
ws = [torch.randn(64, 128, 1, 1, device=dev) * 0.09, torch.randn(64, 64, 3, 3, device=dev) * 0.04, torch.randn(128, 64, 1, 1, device=dev) * 0.12]
packs = [K.pack_conv(w, 0, K.BF16) for w in ws]
packs_d = [K.pack_conv(w, 1, K.BF16) for w in ws]
biases = [torch.randn(c, device=dev) * 0.1 for c in (64, 64, 128)]
gammas = [1 + 0.1 * torch.randn(c, device=dev) for c in (128, 64, 64)]
betas = [0.1 * torch.randn(c, device=dev) for c in (128, 64, 64)]
FWD = ["params + x load + norm a", "GEMM a", "acc -> LDS + halo", "norm b", "GEMM b (3x3)", "acc -> LDS", "norm c", "GEMM c", "acc -> LDS", "residual + store"]
BWD = ["g_out load (+ states)", "GEMM c^T + acc + halo", "norm-bwd c", "GEMM b^T + acc", "norm-bwd b", "GEMM a^T + acc", "norm-bwd a + store"]
for W in (16, 8, 4, 2):
    x = torch.randn(B, W, W, C, device=dev).to(torch.bfloat16)
    gout = torch.randn(B, W, W, C, device=dev).to(torch.bfloat16)
    out, t1, t2, st = K.resblock_fwd_small(x, packs, biases, gammas, betas)
    for name, fn, labels in (("forward", lambda: K.resblock_fwd_small(x, packs, biases, gammas, betas), FWD),
                             ("backward", lambda: K.resblock_bwd_small(gout, x, t1, t2, packs_d, st), BWD)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        stamps = torch.zeros(B, 16, dtype=torch.int64, device=dev)
        l.pwr_debug_set_stamps(stamps.data_ptr())
        fn()
        torch.cuda.synchronize()
        l.pwr_debug_set_stamps(None)
        s = stamps.cpu().double()
        n = len(labels)
        d = (s[:, 1:n + 1] - s[:, 0:n]).mean(dim=0)
        tot = float((s[:, n] - s[:, 0]).mean())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        print("%2dx%-2d %-8s workgroup life %6.0f cycles (s_memtime); us per call incl. host + the parameter-sum launch: %.1f" % (W, W, name, tot, e0.elapsed_time(e1) / 20 * 1e3))
        print("        " + "  |  ".join("%s %.0f" % (lb, float(v)) for lb, v in zip(labels, d)))
