"""Uninitialised-memory screen: the same train steps with the engine arena (and every other buffer the step allocates) pre-filled with
0x00 and with 0xFF (NaN patterns) must give bit-identical gradients and losses."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixelwiseregression_amd import PixelwiseRegression
from pixelwiseregression_amd.synthetic import make_batch
from pixelwiseregression_amd.train import TrainStep
dev = "cuda:0"
def run(fill):
    torch.manual_seed(0)
    m = PixelwiseRegression(14, stage=2, label_size=64, features=128, level=4, norm_method="instance").to(dev).set_precision("bf16").train()
    b = make_batch(32, 14, S=128, seed=1234, device=dev, dense_targets=True)
    ts = TrainStep(m, opt="adam", lr=1e-3, alpha=0.5)
    args = (b["img"], b["label_img"], b["mask"], b["uvd"], b["heatmaps"], b["depthmaps"])
    ts(*args)                                  # builds the plan
    torch.cuda.synchronize()
    for plan in m._engine.values():
        plan.arena.fill_(fill)
    out = []
    for it in range(3):
        ts(*args)
        out.append((m.flat_grad().clone(), ts.loss.clone()))
    torch.cuda.synchronize()
    return out
a, b = run(0x00), run(0xFF)
for it, ((ga, la), (gb, lb)) in enumerate(zip(a, b)):
    d = (ga != gb) & ~(ga.isnan() & gb.isnan())
    print("step", it, "different gradient elements:", int(d.sum()), "NaN in grads:", int(gb.isnan().sum()), "loss", la.item(), lb.item())
