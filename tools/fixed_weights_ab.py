"""Train-step time with FIXED weights (SGD, lr 0) on one batch, debug build: an A/B between two settings of an experiment switch then
runs on (nearly) the same activations step after step, which takes the data-dependent part of the chip's clock management out of the
comparison (two builds that differ in the last bits train to different weights within a few steps on random data, and the dominant conv
alone moves 20 % between random and all-zero operands).      PWR_X=... python tools/fixed_weights_ab.py [steps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dbglib  # noqa: F401
from pixelwiseregression_amd import PixelwiseRegression
from pixelwiseregression_amd.synthetic import make_batch
from pixelwiseregression_amd.train import TrainStep
dev = "cuda:0"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
torch.manual_seed(0)
m = PixelwiseRegression(14, stage=2, label_size=64, features=128, level=4, norm_method="instance").to(dev).set_precision("bf16").train()
b = make_batch(32, 14, S=128, seed=1234, device=dev)
ts = TrainStep(m, opt="sgd", lr=0.0)
args = (b["img"], b["label_img"], b["mask"], b["uvd"])
for _ in range(30): ts(*args)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(N): ts(*args)
e1.record(); torch.cuda.synchronize()
print("%s  %.3f ms/step  loss %.6f" % (" ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("PWR_")), e0.elapsed_time(e1) / N, float(ts.loss)))
