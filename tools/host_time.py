"""Host-side issue time of the native train step vs its GPU time (is the step launch-bound?)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixelwiseregression_amd import PixelwiseRegression
from pixelwiseregression_amd.synthetic import make_batch
from pixelwiseregression_amd.train import TrainStep
dev = "cuda:0"
torch.manual_seed(0)
m = PixelwiseRegression(14, stage=2, label_size=64, features=128, level=4, norm_method="instance").to(dev).train()
m.set_precision("bf16")
b = make_batch(32, 14, S=128, seed=3, device=dev)
ts = TrainStep(m, lr=1e-4)
for _ in range(5): ts(b["img"], b["label_img"], b["mask"], b["uvd"])
torch.cuda.synchronize()
N = 30
t0 = time.perf_counter()
host = 0.0
for _ in range(N):
    h0 = time.perf_counter()
    ts(b["img"], b["label_img"], b["mask"], b["uvd"])
    host += time.perf_counter() - h0
    torch.cuda.synchronize()          # one step at a time: host issue time is not hidden behind the previous step
t1 = time.perf_counter()
print("per step: wall %.3f ms (synchronised each step), host issue %.3f ms" % ((t1 - t0) / N * 1e3, host / N * 1e3))
