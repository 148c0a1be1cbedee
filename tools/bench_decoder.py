"""Decoder micro-benchmark: achieved algorithmic GB/s vs the HBM roofline.

bytes_fwd = 12*B*J*P^2 + 8*B*P^2 + 12*B*J ;  bytes_bwd = 28*B*J*P^2 + 8*B*P^2 (gH and gD given)

Launches go through the C ABI on PREALLOCATED buffers (like bench.py's decoder probe): through ops.decode_forward -- two torch.empty per
call -- the small shapes were host-bound (round 5's file said 17.6 us for the 6.8-us C2 launch).
    python tools/bench_decoder.py [--debug-lib] [shape ...]      shape = B,J,P   (default: the C2, C3 and C5 per-GPU shapes)
With --debug-lib the debug build's switches are live (PWR_DEC_XCD, PWR_DEC_NT, PWR_DEC_FWD128, PWR_DEC_BWD128); they are echoed in the output.
"""
import json
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "--debug-lib" in sys.argv:
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import dbglib  # noqa: F401
import torch  # noqa: E402
from pixelwiseregression_amd import _lib  # noqa: E402


def run(B, J, P, iters=50):
    dev = torch.device("cuda:0")
    l = _lib.lib()
    torch.manual_seed(0)
    z = torch.randn(B, J, P, P, device=dev)
    D = torch.randn(B, J, P, P, device=dev)
    m = (torch.rand(B, 1, P, P, device=dev) < 0.4).float()
    L = torch.randn(B, 1, P, P, device=dev) * m
    w = torch.ones(J, 1, device=dev)
    gH = torch.randn(B, J, P, P, device=dev)
    gD = torch.randn(B, J, P, P, device=dev)
    gU = torch.randn(B, J, 3, device=dev)
    p, uvd = torch.empty_like(z), torch.empty(B, J, 3, device=dev)
    gz, gDo, gw = torch.empty_like(z), torch.empty_like(z), torch.empty(B * J, device=dev)
    st = _lib.stream_ptr(dev)
    P_ = lambda t: t.data_ptr()

    def fwd():
        _lib.check(l.pwr_decode_fwd(P_(z), P_(D), P_(L), P_(m), P_(w), P_(p), P_(uvd), B, J, P, 0, st), "pwr_decode_fwd")

    def bwd():
        _lib.check(l.pwr_decode_bwd(P_(p), P_(z), P_(D), P_(L), P_(m), P_(w), P_(uvd), P_(gH), P_(gD), P_(gU), P_(gz), P_(gDo), P_(gw), B, J, P, 0, st),
                   "pwr_decode_bwd")

    for _ in range(3):
        fwd(); bwd()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fwd()
    e1.record(); torch.cuda.synchronize()
    tf = e0.elapsed_time(e1) / iters * 1e-3
    e0.record()
    for _ in range(iters):
        bwd()
    e1.record(); torch.cuda.synchronize()
    tb = e0.elapsed_time(e1) / iters * 1e-3
    bf = 12 * B * J * P * P + 8 * B * P * P + 12 * B * J
    bb = 28 * B * J * P * P + 8 * B * P * P
    out = {"B": B, "J": J, "P": P, "fwd_us": tf * 1e6, "bwd_us": tb * 1e6, "fwd_GBs": bf / tf / 1e9,
           "bwd_GBs": bb / tb / 1e9, "fwd_frac_8TBs": bf / tf / 8e12, "bwd_frac_8TBs": bb / tb / 8e12,
           "checksum": float(p.double().sum() + gz.double().abs().sum())}
    sw = {k: v for k, v in os.environ.items() if k.startswith("PWR_DEC")}
    if sw:
        out["switches"] = sw
    return out


if __name__ == "__main__":
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:] if "," in a] or [(32, 14, 64), (64, 21, 64), (128, 42, 128)]
    for (B, J, P) in shapes:
        print(json.dumps(run(B, J, P)), flush=True)
