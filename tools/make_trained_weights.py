"""GPU side of the TRAINED fixture (tests/golden/trained_c2.npz): train BASELINE C2's architecture for a while on the stream of rendered
synthetic hands (evaluate.train_and_validate's loop, /root/reference/train.py:158-212), then save the state_dict and a few held-out
frames to gpurun_out/trained_c2_weights.npz.  oracle/gen_trained_golden.py (build container, reference importable) turns that into the
fixture by running the REFERENCE on the saved weights and inputs in float64.

    python tools/make_trained_weights.py [steps] [precision]
"""
import os, sys, json
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pixelwiseregression_amd import PixelwiseRegression
from pixelwiseregression_amd.synthetic import make_pose_batch
from pixelwiseregression_amd.train import TrainStep
from pixelwiseregression_amd.evaluate import validate

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
dev = "cuda:0"
J, S, B = 14, 128, 32
torch.manual_seed(0)
m = PixelwiseRegression(J, stage=2, label_size=S // 2, features=128, level=4, norm_method="instance").to(dev).set_precision(prec).train()
ts = TrainStep(m, opt="adam", lr=1e-3, alpha=1.0)
val = [make_pose_batch(B, J, S, seed=10_000_000 + k, device=dev) for k in range(2)]
e0, _ = validate(m, val)
for it in range(steps):
    if it == int(steps * 0.7):
        ts.lr *= 0.2                      # one StepLR decay (train.py:143): ends in a flatter region
    b = make_pose_batch(B, J, S, seed=it + 1, device=dev)
    ts(b["img"], b["label_img"], b["mask"], b["uvd"])
e1, _ = validate(m, val)
print(json.dumps({"steps": steps, "precision": prec, "mm_untrained": e0, "mm_trained": e1}))
held = make_pose_batch(4, J, S, seed=20_000_000, device=dev)
rec = {"sd_" + k: v.detach().float().cpu().numpy() for k, v in m.state_dict().items()}
for k in ("img", "label_img", "mask", "uvd", "box_size", "cube_size", "com"):
    rec["in_" + k] = held[k].detach().float().cpu().numpy()
rec["mm_trained"] = np.asarray(e1)
rec["train_steps"] = np.int64(steps)
out = os.path.join(ROOT, "gpurun_out")
os.makedirs(out, exist_ok=True)
np.savez_compressed(os.path.join(out, "trained_c2_weights.npz"), **rec)
# information only: what the two engines produce on the held-out frames (the fixture's expectations come from the reference in float64)
m.eval()
info = {}
with torch.no_grad():
    for p in ("fp32", "bf16"):
        m.set_precision(p)
        res = m(held["img"], held["label_img"], held["mask"])
        info[p] = [r[2].cpu().numpy() for r in res]
print("bf16 vs fp32 engine uvd, held-out:", [float(np.abs(a - b_).max()) for a, b_ in zip(info["fp32"], info["bf16"])])
