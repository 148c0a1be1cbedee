"""Is the train step host-bound?  Time how long the HOST needs to issue K steps (the loop returns) against when the GPU finishes them."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixelwiseregression_amd import PixelwiseRegression
from pixelwiseregression_amd.synthetic import make_batch
from pixelwiseregression_amd.train import TrainStep
dev = "cuda:0"
torch.manual_seed(0)
m = PixelwiseRegression(14, stage=2, label_size=64, features=128, level=4, norm_method="instance").to(dev).set_precision("bf16").train()
tr = TrainStep(m, opt="adam", lr=1e-4, beta1=0.9, beta2=0.999, weight_decay=0.0, alpha=1.0, lambda_h=1.0, lambda_d=0.01)
b = make_batch(32, 14, S=128, seed=1234, device=dev, dense_targets=True)
step = lambda: tr(b["img"], b["label_img"], b["mask"], b["uvd"], b["heatmaps"], b["depthmaps"])
for _ in range(20): step()
torch.cuda.synchronize()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 100
t0 = time.perf_counter()
for _ in range(K): step()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("issue %.3f ms/step, complete %.3f ms/step (host-bound if equal)" % (t_issue / K * 1e3, t_all / K * 1e3))
# inference
m.eval()
with torch.no_grad():
    for _ in range(10): m(b["img"], b["label_img"], b["mask"])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K): m(b["img"], b["label_img"], b["mask"])
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize(); t_all = time.perf_counter() - t0
print("inference: issue %.3f ms, complete %.3f ms per forward" % (t_issue / K * 1e3, t_all / K * 1e3))
