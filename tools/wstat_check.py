"""conv3x3_wstat_kernel (conv_wstat.hip) against conv3x3_patch_kernel (conv_patch.hip) on the same inputs, in ONE process through the debug
build's PWR_WSTAT switch: outputs and epilogue statistics must be bit-identical in every (norm prologue, statistics kind) form; then
interleaved timing of both.    python tools/wstat_check.py [B]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dbglib  # noqa: F401
import torch
from pixelwiseregression_amd import kernels as K
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import kat_cases as kc

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = "cuda:0"


def run(form, which):
    os.environ["PWR_WSTAT"] = str(which)
    nrm, kind = form
    x = kc.dev((B, 64, 64, 128), 1, dtype=torch.bfloat16)
    pack = K.pack_conv(kc.det((128, 128, 3, 3), 2, 0.05).to(dev), 1 if kind == 2 else 0, K.BF16)
    bias = kc.dev((128,), 3, 0.5) if kind != 2 else None
    st = kc._state(B, 128, 4) if nrm else None
    if kind == 0:
        y, _ = K.conv_fwd(x, pack, 128, 3, 1, bias=bias, norm=st)
        return [y]
    if kind == 1:
        y, part, _ = K.conv_fwd_stats(x, pack, 128, 3, 1, bias=bias, norm=st)
        return [y, part]
    nby = kc.dev((B, 64, 64, 128), 5, dtype=torch.bfloat16)
    y, part, _ = K.conv_fwd_stats(x, pack, 128, 3, 1, norm=st, nb_y=nby, nb_state=kc._state(B, 128, 6))
    return [y, part]


out = {}
for nrm in (0, 1):
    for kind in (0, 1, 2):
        a, b = run((nrm, kind), 0), run((nrm, kind), 1)
        torch.cuda.synchronize()
        diffs = []
        for ta, tb in zip(a, b):
            ne = (ta.float() != tb.float()) & ~(torch.isnan(ta.float()) & torch.isnan(tb.float()))
            diffs.append({"differ": int(ne.sum()), "of": ta.numel(), "max_abs": float((ta.float() - tb.float()).abs().max())})
            if int(ne.sum()):
                idx = ne.nonzero()[:6].tolist()
                diffs[-1]["first"] = idx
        out["nrm%d_kind%d" % (nrm, kind)] = diffs
print(json.dumps(out))
# timing, interleaved
res = {}
for form in ((1, 0), (1, 1), (0, 2)):
    ts = {0: [], 1: []}
    for rep in range(5):
        for which in (0, 1):
            os.environ["PWR_WSTAT"] = str(which)
            run(form, which); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            nrm, kind = form
            x = kc.dev((B, 64, 64, 128), 1, dtype=torch.bfloat16)
            pack = K.pack_conv(kc.det((128, 128, 3, 3), 2, 0.05).to(dev), 1 if kind == 2 else 0, K.BF16)
            st = kc._state(B, 128, 4) if nrm else None
            nby = kc.dev((B, 64, 64, 128), 5, dtype=torch.bfloat16); nbs = kc._state(B, 128, 6)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(20):
                if kind == 0: K.conv_fwd(x, pack, 128, 3, 1, norm=st)
                elif kind == 1: K.conv_fwd_stats(x, pack, 128, 3, 1, norm=st)
                else: K.conv_fwd_stats(x, pack, 128, 3, 1, nb_y=nby, nb_state=nbs)
            e1.record(); torch.cuda.synchronize()
            ts[which].append(e0.elapsed_time(e1) / 20 * 1e3)
    res["nrm%d_kind%d" % form] = {"patch_us": [round(v, 1) for v in ts[0]], "wstat_us": [round(v, 1) for v in ts[1]]}
print(json.dumps(res))
