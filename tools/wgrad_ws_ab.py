"""A/B of the wave-specialised 128-channel weight gradient (csrc/conv_wgrad_ws.hip) against the kernels it replaces, in ONE process on the
DEBUG build (PWR_WGRAD3W=0 / 1 is read per call there): bit-identical dW at equal split counts, and kernel + reduce time by HIP events,
interleaved rounds (cdna_hip_programming.md rule 24).  C2 heads shape: B=32, 64x64, 128 -> 128.

    python tools/build_debug.py && python tools/wgrad_ws_ab.py
"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dbglib  # noqa: F401
import torch
from pixelwiseregression_amd import kernels as K

dev = "cuda:0"
torch.manual_seed(0)


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def run(B, H, W, Cin, Cout, splits, norm):
    x = torch.randn(B, H, W, Cin, device=dev).to(torch.bfloat16)
    dy = torch.randn(B, H, W, Cout, device=dev).to(torch.bfloat16)
    st = K.norm_stats(x, 1 + 0.3 * torch.randn(Cin, device=dev), 0.3 * torch.randn(Cin, device=dev), mode=0) if norm else None
    out = {}
    modes = {"r3": ("0", "0"), "ws3": ("1", "0"), "ws9": ("1", "1")}        # (PWR_WGRAD3W, PWR_WGRAD9W): round-3 kernels, three-tap, nine-tap wave-specialised
    def setmode(m):
        os.environ["PWR_WGRAD3W"], os.environ["PWR_WGRAD9W"] = modes[m]
    for m in modes:
        setmode(m)
        out[m] = K.conv_wgrad(x, dy, Cout, 3, 1, norm=st, relu_in=True, splits=splits).clone()
    t = {m: [] for m in modes}
    for _ in range(3):
        for m in modes:
            setmode(m)
            t[m].append(timeit(lambda: K.conv_wgrad(x, dy, Cout, 3, 1, norm=st, relu_in=True, splits=splits)))
    flops = 2.0 * B * H * W * Cin * Cout * 9
    scale = float(out["r3"].abs().max())
    rec = {"shape": [B, H, W, Cin, Cout], "splits": splits, "norm": norm, "ws3_bit_identical_to_r3": bool(torch.equal(out["r3"], out["ws3"])),
           "ws9_max_rel_diff_to_r3": float((out["r3"] - out["ws9"]).abs().max()) / scale,
           "r3_us": [round(v, 1) for v in t["r3"]], "ws3_us": [round(v, 1) for v in t["ws3"]], "ws9_us": [round(v, 1) for v in t["ws9"]],
           "ws9_TFLOPs_incl_reduce": flops / (min(t["ws9"]) * 1e-6) / 1e12}
    print(json.dumps(rec), flush=True)
    return x, dy, st


for norm in (True, False):
    for splits in (80, 24):
        x, dy, st = run(32, 64, 64, 128, 128, splits, norm)
    continue
    # the pair launch: two layers, half the splits each
    os.environ["PWR_WGRAD3W"] = "1"
    x2 = torch.randn_like(x.float()).to(torch.bfloat16)
    dy2 = torch.randn_like(dy.float()).to(torch.bfloat16)
    st2 = K.norm_stats(x2, torch.ones(128, device=dev), torch.zeros(128, device=dev), mode=0) if norm else None
    for sp in (40, 80):
        t = min(timeit(lambda: K.conv_wgrad_pair(x, dy, x2, dy2, norm_a=st, norm_b=st2, splits=sp)) for _ in range(3))
        print(json.dumps({"pair": True, "norm": norm, "splits_each": sp, "us_per_pair_incl_reduce": round(t, 1), "us_per_layer": round(t / 2, 1)}), flush=True)
run(8, 128, 128, 128, 128, 85, True)
run(3, 20, 96, 128, 128, 24, True)
