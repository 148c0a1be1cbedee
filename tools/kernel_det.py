"""One kernel, N launches on the same inputs, outputs compared bitwise with the first launch ON THE DEVICE (no host sync in the loop);
the first differing output is kept and analysed.

    python tools/kernel_det.py [launches] [which]      which: nb (data gradient + norm-backward sums), st (forward + statistics), plain
"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixelwiseregression_amd import kernels as K
dev = "cuda:0"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
which = sys.argv[2] if len(sys.argv) > 2 else "nb"
torch.manual_seed(0)
bf = torch.bfloat16
B = 32
x = torch.randn(B, 64, 64, 128, device=dev).to(bf)
w = torch.randn(128, 128, 3, 3, device=dev) * 0.03
pf, pd = K.pack_conv(w, 0, K.BF16), K.pack_conv(w, 1, K.BF16)
st = K.norm_stats(x, torch.ones(128, device=dev), torch.zeros(128, device=dev))
y = torch.randn(B, 64, 64, 128, device=dev).to(bf)
bias = torch.randn(128, device=dev) * 0.1
if which == "nb":
    fn = lambda: K.conv_fwd_stats(x, pd, 128, 3, 1, nb_y=y, nb_state=st)[:2]
elif which == "st":
    fn = lambda: K.conv_fwd_stats(x, pf, 128, 3, 1, bias=bias, norm=st)[:2]
else:
    fn = lambda: (K.conv_fwd(x, pf, 128, 3, 1, bias=bias, norm=st)[0], torch.zeros(1, device=dev))
o0, p0 = [t.clone() for t in fn()]
cap_o, cap_p = torch.zeros_like(o0), torch.zeros_like(p0)
have = torch.zeros((), dtype=torch.bool, device=dev)
bad_o = torch.zeros((), device=dev); bad_p = torch.zeros((), device=dev)
torch.cuda.synchronize()
t0 = time.time()
for it in range(N):
    o, p = fn()
    fo = (o != o0).any(); fp = ((p != p0) & ~(p.isnan() & p0.isnan())).any()
    bad_o += fo; bad_p += fp
    take = (fo | fp) & ~have
    cap_o = torch.where(take, o, cap_o); cap_p = torch.where(take, p, cap_p)
    have |= take
torch.cuda.synchronize()
print("%s: %d launches, %.1f us each (with the checks); different output tensor: %d, different statistics: %d"
      % (which, N, (time.time() - t0) / N * 1e6, int(bad_o), int(bad_p)))
if bool(have):
    d = (cap_o != o0).nonzero()
    print("  output elements that differ:", d.shape[0])
    if d.shape[0]:
        print("    samples", d[:, 0].unique().tolist()[:8], "rows", d[:, 1].unique().tolist()[:12], "cols", d[:, 2].unique().tolist()[:12],
              "channels", d[:, 3].unique().tolist()[:16])
        i = tuple(d[0].tolist()); print("    e.g.", i, float(cap_o[i]), "vs", float(o0[i]))
    dp = ((cap_p != p0) & ~(cap_p.isnan() & p0.isnan())).nonzero()
    print("  statistics entries that differ:", dp.shape[0], dp[:8].tolist())
    for i in dp[:4].tolist():
        i = tuple(i); print("    ", i, float(cap_p[i]), "vs", float(p0[i]))
