"""Launch-to-launch bitwise stability of the kernels added late in round 3, N launches each on fixed inputs, compared with the first
launch on the device: the stride-2 stem conv (parity-class form, with its statistics), the LDS-DMA weight gradient with the norm
applied in LDS, the small-map ResBlock with a fused max-pool / up-sample input.    python tools/new_kernels_det.py [launches]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixelwiseregression_amd import kernels as K
dev = "cuda:0"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
bf = torch.bfloat16
torch.manual_seed(0)

def soak(name, fn):
    ref = [t.clone() for t in fn()]
    bad = torch.zeros((), device=dev)
    for _ in range(N):
        out = fn()
        for a, b in zip(out, ref):
            bad += ((a != b) & ~(a.isnan() & b.isnan())).any()
    torch.cuda.synchronize()
    print("%-60s %d launches, %d with a different output" % (name, N, int(bad)))

B = 32
x = torch.randn(B, 128, 128, 128, device=dev).to(bf)
w = torch.randn(128, 128, 3, 3, device=dev) * 0.03
pf = K.pack_conv(w, 0, K.BF16)
st = K.norm_stats(x, torch.ones(128, device=dev), torch.zeros(128, device=dev))
bias = torch.randn(128, device=dev) * 0.1
soak("stride-2 conv 128->128, 128x128, norm prologue + statistics", lambda: K.conv_fwd_stats(x, pf, 128, 3, 2, bias=bias, norm=st)[:2])
del x
x = torch.randn(B, 64, 64, 64, device=dev).to(bf); dy = torch.randn(B, 64, 64, 64, device=dev).to(bf)
st = K.norm_stats(x, 1 + 0.2 * torch.randn(64, device=dev), 0.2 * torch.randn(64, device=dev))
soak("3x3 weight gradient 64->64, 64x64, norm in LDS, 80 splits", lambda: (K.conv_wgrad(x, dy, 64, 3, 1, norm=st, relu_in=True, splits=80),))
C, Fh = 128, 64
ws = [torch.randn(Fh, C, 1, 1, device=dev) * C ** -0.5, torch.randn(Fh, Fh, 3, 3, device=dev) * (9 * Fh) ** -0.5, torch.randn(C, Fh, 1, 1, device=dev) * Fh ** -0.5]
wf = [K.pack_conv(w_, 0, K.BF16) for w_ in ws]
bs = [torch.randn(c, device=dev) * 0.1 for c in (Fh, Fh, C)]
gs = [1 + 0.2 * torch.randn(c, device=dev) for c in (C, Fh, Fh)]
bes = [0.2 * torch.randn(c, device=dev) for c in (C, Fh, Fh)]
a32 = torch.randn(B, 32, 32, C, device=dev).to(bf); a16 = torch.randn(B, 16, 16, C, device=dev).to(bf); h8 = torch.randn(B, 8, 8, C, device=dev).to(bf)
soak("small ResBlock 16x16, input = maxpool(32x32)", lambda: K.resblock_fwd_small_x(1, a32, None, wf, bs, gs, bes)[:4])
soak("small ResBlock 16x16, input = upsample(8x8) + skip", lambda: K.resblock_fwd_small_x(2, a16, h8, wf, bs, gs, bes)[:4])
