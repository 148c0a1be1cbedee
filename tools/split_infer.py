"""Experiment: inference of a batch as TWO half batches on two streams (the small-map kernels of one half overlap the big
convolutions of the other) vs one full batch on one stream."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixelwiseregression_amd import PixelwiseRegression
from pixelwiseregression_amd.synthetic import make_batch
dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
torch.manual_seed(0)
m = PixelwiseRegression(14, stage=2, label_size=64, features=128, level=4, norm_method="instance").to(dev).set_precision("bf16").eval()
b = make_batch(B, 14, S=128, seed=1, device=dev)
halves = [{k: v[i * B // 2:(i + 1) * B // 2].contiguous() for k, v in b.items()} for i in range(2)]
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
def full():
    return m(b["img"], b["label_img"], b["mask"])
def split():
    outs = []
    cur = torch.cuda.current_stream(dev)
    for s, h in zip(streams, halves):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            outs.append(m(h["img"], h["label_img"], h["mask"]))
    for s in streams:
        cur.wait_stream(s)
    return outs
with torch.no_grad():
    for fn, name in ((full, "one batch of %d" % B), (split, "two halves on two streams"), (full, "one batch again")):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 100
        for _ in range(n): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        print("%-28s %.3f ms  %.0f frames/s" % (name, dt * 1e3, B / dt))
    r = full(); h = split()
    print("max |diff| uvd:", max((r[-1][2][:B // 2] - h[0][-1][2]).abs().max().item(), (r[-1][2][B // 2:] - h[1][-1][2]).abs().max().item()))
