"""Inference only (forward + decode, no_grad) at BASELINE C2's shape, for rocprofv3 --stats."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixelwiseregression_amd import PixelwiseRegression
from pixelwiseregression_amd.synthetic import make_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
torch.manual_seed(0)
m = PixelwiseRegression(14, stage=2, label_size=64, features=128, level=4, norm_method="instance").to("cuda:0").set_precision("bf16").eval()
b = make_batch(B, 14, S=128, seed=1, device="cuda:0")
with torch.no_grad():
    for _ in range(5): m(b["img"], b["label_img"], b["mask"])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): m(b["img"], b["label_img"], b["mask"])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print("B=%d: %.3f ms per forward, %.0f frames/s" % (B, dt * 1e3, B / dt))
