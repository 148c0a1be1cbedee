import os, sys, copy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixelwiseregression_amd import PixelwiseRegression
from pixelwiseregression_amd.synthetic import make_batch
from pixelwiseregression_amd.train import TrainStep
dev = "cuda:0"
torch.manual_seed(0)
m = PixelwiseRegression(14, stage=2, label_size=64, features=128, level=4, norm_method="instance").to(dev).train()
m.set_precision("bf16")
b = make_batch(8, 14, S=128, seed=3, device=dev)
ts = TrainStep(m, lr=1e-4)
sync = len(sys.argv) > 1 and sys.argv[1] == "sync"
losses = []
for it in range(12):
    l = ts(b["img"], b["label_img"], b["mask"], b["uvd"])
    if sync: torch.cuda.synchronize()
    losses.append(l.clone())
torch.cuda.synchronize()
print(["%.5f" % x.item() for x in losses])
