"""Ping-pong 3x3 conv vs the one-tile-per-workgroup kernel (same process, pwr_debug_set_pingpong): bitwise comparison over
several shapes (ragged tile counts included) and repeats, then timing."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixelwiseregression_amd import kernels as K, _lib
l = _lib.lib()
dev = "cuda:0"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
bad = 0
for B, H, W in ((32, 64, 64), (33, 64, 64), (48, 64, 64), (8, 128, 128), (5, 128, 160), (64, 64, 64)):
    torch.manual_seed(B)
    x = torch.randn(B, H, W, 128, device=dev).to(torch.bfloat16)
    w = torch.randn(128, 128, 3, 3, device=dev) * 0.03
    pf, pd = K.pack_conv(w, 0, K.BF16), K.pack_conv(w, 1, K.BF16)
    st = K.norm_stats(x, torch.rand(128, device=dev) + 0.5, torch.randn(128, device=dev) * 0.1)
    bias = torch.randn(128, device=dev) * 0.1
    yv = torch.randn(B, H, W, 128, device=dev).to(torch.bfloat16)
    fns = {"fwd": lambda: (K.conv_fwd(x, pf, 128, 3, 1, bias=bias, norm=st)[0],),
           "plain": lambda: (K.conv_fwd(x, pd, 128, 3, 1)[0],),
           "st": lambda: K.conv_fwd_stats(x, pf, 128, 3, 1, bias=bias, norm=st)[:2],
           "nb": lambda: K.conv_fwd_stats(x, pd, 128, 3, 1, nb_y=yv, nb_state=st)[:2]}
    for name, fn in fns.items():
        l.pwr_debug_set_pingpong(0)
        ref = [t.clone() for t in fn()]
        l.pwr_debug_set_pingpong(1)
        nbad = 0
        for _ in range(reps):
            out = fn()
            for a, b in zip(out, ref):
                if not torch.equal(torch.nan_to_num(a.float(), nan=777.0), torch.nan_to_num(b.float(), nan=777.0)):
                    nbad += 1
                    d = (torch.nan_to_num(a.float()) - torch.nan_to_num(b.float())).abs()
                    if nbad <= 2: print("   DIFFERENT %dx%dx%d %s: %d elements, max %.3g, first %s" % (B, H, W, name, int((d > 0).sum()), d.max().item(), (d > 0).nonzero()[:2].tolist()))
        bad += nbad
        print("%dx%dx%d %-6s %s" % (B, H, W, name, "ok" if nbad == 0 else "%d of %d comparisons differ" % (nbad, reps * len(ref))))
    if (B, H) in ((32, 64), (64, 64)):
        for mode in (0, 1):
            l.pwr_debug_set_pingpong(mode)
            for name, fn in fns.items():
                for _ in range(5): fn()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(200): fn()
                torch.cuda.synchronize(); print("   pingpong=%d %-6s %.1f us" % (mode, name, (time.perf_counter() - t0) / 200 * 1e6))
l.pwr_debug_set_pingpong(-1)
print("TOTAL different:", bad)
