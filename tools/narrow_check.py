"""The narrow form of conv3x3_wstat_kernel (KIND 3: 128 -> J <= 32 channels, fp32 NCHW output -- the heads' last conv) against
conv3x3_patch_kernel<bf16, 128, 4, 1, 1, 1> on the same inputs, in ONE process through the debug build's PWR_WSTAT_NARROW switch: outputs must
be bit-identical; then some launches of both for a rocprofv3 --kernel-trace --stats run around this script.   python tools/narrow_check.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dbglib  # noqa: F401
import torch
from pixelwiseregression_amd import kernels as K
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import kat_cases as kc

dev = "cuda:0"


def run(B, H, W, J, nrm, bias, which):
    os.environ["PWR_WSTAT_NARROW"] = str(which)
    x = kc.dev((B, H, W, 128), 1, dtype=torch.bfloat16)
    pack = K.pack_conv(kc.det((J, 128, 3, 3), 2, 0.05).to(dev), 0, K.BF16)
    _, yn = K.conv_fwd(x, pack, J, 3, 1, bias=kc.dev((J,), 3, 0.5) if bias else None, norm=kc._state(B, 128, 4) if nrm else None,
                       nhwc_out=False, nchw_out=True)
    return yn


out = {}
for (B, H, W, J) in ((32, 64, 64, 14), (2, 64, 64, 21), (3, 32, 64, 32), (1, 64, 32, 1), (5, 36, 96, 14), (32, 64, 64, 21)):
    for nrm in (0, 1):
        for bias in (0, 1):
            a, b = run(B, H, W, J, nrm, bias, 0), run(B, H, W, J, nrm, bias, 1)
            torch.cuda.synchronize()
            ne = (a != b) & ~(torch.isnan(a) & torch.isnan(b))
            out["B%d_%dx%d_J%d_nrm%d_bias%d" % (B, H, W, J, nrm, bias)] = {"differ": int(ne.sum()), "of": a.numel(), "max_abs": float((a - b).abs().max())}
# the 64-input-channel form of the wide kernel (the stem's 64 -> 128 conv) against conv3x3_patch_kernel<bf16, 64, ...>: PWR_WSTAT_C64
def run64(B, H, W, nrm, stats, which):
    os.environ["PWR_WSTAT_C64"] = str(which)
    x = kc.dev((B, H, W, 64), 11, dtype=torch.bfloat16)
    pack = K.pack_conv(kc.det((128, 64, 3, 3), 12, 0.05).to(dev), 0, K.BF16)
    bias = kc.dev((128,), 13, 0.5)
    st = kc._state(B, 64, 14) if nrm else None
    if stats:
        y, part, _ = K.conv_fwd_stats(x, pack, 128, 3, 1, bias=bias, norm=st)
        return [y, part]
    y, _ = K.conv_fwd(x, pack, 128, 3, 1, bias=bias, norm=st)
    return [y]


for (B, H, W) in ((32, 128, 128), (3, 36, 96), (2, 64, 64)):
    for nrm in (0, 1):
        for stats in (0, 1):
            a, b = run64(B, H, W, nrm, stats, 0), run64(B, H, W, nrm, stats, 1)
            torch.cuda.synchronize()
            for i, (ta, tb) in enumerate(zip(a, b)):
                ne = (ta.float() != tb.float()) & ~(torch.isnan(ta.float()) & torch.isnan(tb.float()))
                out["c64_B%d_%dx%d_nrm%d_stats%d_out%d" % (B, H, W, nrm, stats, i)] = {"differ": int(ne.sum()), "of": ta.numel(), "max_abs": float((ta.float() - tb.float()).abs().max())}
print(json.dumps(out))
print("ALL IDENTICAL" if all(v["differ"] == 0 for v in out.values()) else "DIFFERENT")
for which in (0, 1):
    for _ in range(20):
        run64(32, 128, 128, 1, 1, which)
for which in (0, 1):
    for _ in range(30):
        run(32, 64, 64, 14, 1, 1, which)
torch.cuda.synchronize()
