"""The decoder backward ALONE, bit-compared launch by launch, while MFMA weight-gradient kernels run beside it on a second
(lowest-priority) stream -- the condition under which round 1's rare non-reproducible train step arose.

    python tools/kernel_race.py [iterations] [--no-side]

Prints the number of launches whose gz / gDt / gw_part differed from the first launch's.  With the library built WITH the SLP
vectoriser (compiler-packed v_pk_add_f32 ... op_sel forms in decode_bwd_cached) affected boxes show events; the product build
(-fno-slp-vectorize) must show none.  tests/test_decoder_gpu.py runs a short version of this as a regression test.
"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(iters=60000, side=True, B=32, J=14, P=64, dev="cuda:0", wgrad_every=6, verbose=False):
    from pixelwiseregression_amd import _lib, kernels as K
    l = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(5)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    z, D = r(B, J, P, P), r(B, J, P, P)
    m = (torch.rand(B, 1, P, P, generator=g) < 0.3).float().to(dev)
    L = r(B, 1, P, P) * m
    w = (1 + 0.3 * torch.randn(J, generator=g)).to(dev)
    gH, gD, gU = r(B, J, P, P) * 1e-4, r(B, J, P, P) * 1e-4, r(B, J, 3) * 1e-3
    p = torch.empty_like(z); uvd = torch.empty(B, J, 3, device=dev)
    s_main = torch.cuda.current_stream(dev)
    sp = s_main.cuda_stream
    _lib.check(l.pwr_decode_fwd(z.data_ptr(), D.data_ptr(), L.data_ptr(), m.data_ptr(), w.data_ptr(), p.data_ptr(), uvd.data_ptr(), B, J, P, 0, sp), "fwd")
    gz, gDt, gwp = torch.empty_like(z), torch.empty_like(z), torch.empty(B * J, device=dev)

    def bwd():
        _lib.check(l.pwr_decode_bwd(p.data_ptr(), z.data_ptr(), D.data_ptr(), L.data_ptr(), m.data_ptr(), w.data_ptr(), uvd.data_ptr(), gH.data_ptr(),
                                    gD.data_ptr(), gU.data_ptr(), gz.data_ptr(), gDt.data_ptr(), gwp.data_ptr(), B, J, P, 0, sp), "bwd")
    bwd()
    gz0, gDt0, gwp0 = gz.clone(), gDt.clone(), gwp.clone()
    # the co-runner: the 3x3 128->128 weight gradient of the heads (bf16 MFMA), on a lowest-priority stream like the engine's
    lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
    s_side = torch.cuda.Stream(dev, priority=0)
    x = torch.randn(B, P, P, 128, device=dev).to(torch.bfloat16)
    dy = (torch.randn(B, P, P, 128, device=dev) * 0.01).to(torch.bfloat16)
    st = K.norm_stats(x, torch.ones(128, device=dev), torch.zeros(128, device=dev), mode=0)
    dw = None
    bad = torch.zeros(3, device=dev)
    first_bad = torch.full((1,), -1.0, device=dev)
    torch.cuda.synchronize()
    t0 = time.time()
    for it in range(iters):
        if side and it % wgrad_every == 0:
            with torch.cuda.stream(s_side):
                dw = K.conv_wgrad(x, dy, 128, 3, 1, norm=st, splits=8, dw=dw)
        bwd()
        e = torch.stack([(gz != gz0).any(), (gDt != gDt0).any(), (gwp != gwp0).any()]).float()
        bad += e
    torch.cuda.synchronize()
    dt = time.time() - t0
    res = {"iters": iters, "side": side, "bad_gz": int(bad[0]), "bad_gDt": int(bad[1]), "bad_gw": int(bad[2]), "us_per_iter": dt / iters * 1e6}
    if verbose:
        print(res)
    return res


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 60000
    print(run(n, side="--no-side" not in sys.argv))
