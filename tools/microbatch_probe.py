"""Feasibility probe: inference of the BASELINE C2 batch (32) as ONE forward against TWO half-batch forwards on two streams (two module
replicas with the same weights, one plan / arena each): do the latency-bound hourglass kernels of one half overlap the other half's
full-chip convs?   python tools/microbatch_probe.py"""
import os, sys, time, json, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pixelwiseregression_amd import PixelwiseRegression
from pixelwiseregression_amd.synthetic import make_batch

dev = "cuda:0"
torch.manual_seed(0)
m = PixelwiseRegression(14, stage=2, label_size=64, features=128, level=4, norm_method="instance").to(dev).set_precision("bf16").eval()
m2 = copy.deepcopy(m)
m3 = copy.deepcopy(m)
m4 = copy.deepcopy(m)
b = make_batch(32, 14, S=128, seed=1, device=dev)
args = (b["img"], b["label_img"], b["mask"])


def split(n):
    k = 32 // n
    return [tuple(t[i * k:(i + 1) * k].contiguous() for t in args) for i in range(n)]


def bench(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


with torch.no_grad():
    t_one = bench(lambda: m(*args))
    res = {"one forward, B=32": round(t_one, 3)}
    for n, mods in ((2, [m, m2]), (4, [m, m2, m3, m4])):
        parts = split(n)
        streams = [torch.cuda.Stream() for _ in range(n)]

        def multi():
            cur = torch.cuda.current_stream()
            for s in streams: s.wait_stream(cur)
            for mod, part, s in zip(mods, parts, streams):
                with torch.cuda.stream(s):
                    mod(*part)
            for s in streams: cur.wait_stream(s)
        res["%d forwards of B=%d on %d streams" % (n, 32 // n, n)] = round(bench(multi), 3)
        # the same forwards one after the other on one stream
        res["%d forwards of B=%d, one stream" % (n, 32 // n)] = round(bench(lambda: [mod(*part) for mod, part in zip(mods, parts)]), 3)
    # check: outputs of the halves equal the full batch's rows
    full = m(*args)
    h0 = m2(*split(2)[0])
    res["max |uvd(full)[:16] - uvd(half)|"] = float((full[-1][2][:16] - h0[-1][2]).abs().max())
print(json.dumps(res))
