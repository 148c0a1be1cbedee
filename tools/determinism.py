"""Run-to-run determinism of the native train step: N identical steps (SGD with lr 0: the weights never change), every step's flat
gradient is compared with step 0's ON THE DEVICE (no host synchronisation inside the loop); afterwards: which steps / parameters differed.

    python tools/determinism.py [steps] [alpha]
"""
import os, sys, time, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixelwiseregression_amd import PixelwiseRegression
from pixelwiseregression_amd.synthetic import make_batch
from pixelwiseregression_amd.train import TrainStep
dev = "cuda:0"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
alpha = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
torch.manual_seed(0)
m = PixelwiseRegression(14, stage=2, label_size=64, features=128, level=4, norm_method="instance").to(dev).set_precision("bf16").train()
b = make_batch(32, 14, S=128, seed=1234, device=dev, dense_targets=alpha != 1.0)
ts = TrainStep(m, opt="sgd", lr=0.0, alpha=alpha)
names = list(m._offsets.keys())
n = m.flat_grad().numel()
pid = torch.empty(n, dtype=torch.long, device=dev)
for i, name in enumerate(names):
    o, shape = m._offsets[name]
    pid[o:o + int(torch.Size(shape).numel())] = i
args = (b["img"], b["label_img"], b["mask"], b["uvd"]) + ((b["heatmaps"], b["depthmaps"]) if alpha != 1.0 else ())
ts(*args)
g0 = m.flat_grad().clone()
l0 = ts.loss.clone()
counts = torch.zeros(N, len(names), device=dev)
lossbad = torch.zeros(N, device=dev)
maxd = torch.zeros(N, device=dev)
torch.cuda.synchronize()
t0 = time.time()
for it in range(N):
    ts(*args)
    g = m.flat_grad()
    d = (g != g0)
    counts[it].index_add_(0, pid, d.float())
    maxd[it] = (g - g0).abs().max()
    lossbad[it] = (ts.loss != l0).float().sum()
torch.cuda.synchronize()
dt = time.time() - t0
bad = (counts.sum(1) > 0).nonzero().flatten().tolist()
print("steps %d  %.2f ms/step  steps with a different gradient: %d %s  different loss: %d" % (N, dt / N * 1e3, len(bad), bad[:20], int(lossbad.sum())))
for it in bad[:6]:
    row = counts[it]
    nz = row.nonzero().flatten().tolist()
    print("  step %d: %d parameters, %d elements, max |diff| %.3e" % (it, len(nz), int(row.sum()), maxd[it].item()))
    # the parameters that differ in FEW elements tell where the divergence started
    few = sorted(nz, key=lambda i: row[i].item())[:8]
    for i in few:
        print("      %-60s %d / %d" % (names[i], int(row[i]), int(torch.Size(m._offsets[names[i]][1]).numel())))
