"""A/B yardstick (NOT the product): BASELINE C2's train step with the conv stack through PyTorch-ROCm library ops (MIOpen / ATen, bf16
autocast; tests/aten_reference.py) + the HIP decoder, autograd + fused torch.optim.AdamW, on the same GPU and the same synthetic batch as
bench.py.  Prints one JSON line.

    python tools/bench_aten.py [steps]
"""
import json, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from aten_reference import aten_forward
from pixelwiseregression_amd import PixelwiseRegression
from pixelwiseregression_amd.synthetic import make_batch
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = "cuda:0"
torch.manual_seed(0)
m = PixelwiseRegression(14, stage=2, label_size=64, features=128, level=4, norm_method="instance").to(dev).set_precision("bf16").train()
opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=0)
b = make_batch(32, 14, S=128, seed=1234, device=dev)


def step():
    opt.zero_grad(set_to_none=True)
    res = aten_forward(m, b["img"], b["label_img"], b["mask"])
    loss = sum(torch.mean(torch.sum((u - b["uvd"]) ** 2, dim=2)) for (_, _, u) in res)
    loss.backward()
    opt.step()
    return loss


for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): loss = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
m.eval()
with torch.no_grad():
    for _ in range(3): aten_forward(m, b["img"], b["label_img"], b["mask"])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): aten_forward(m, b["img"], b["label_img"], b["mask"])
    torch.cuda.synchronize(); di = (time.perf_counter() - t0) / steps
print(json.dumps({"what": "PyTorch-ROCm library convs (MIOpen, bf16 autocast) + HIP decoder, autograd + torch.optim.AdamW; BASELINE C2, B=32",
                  "train_ms_per_step": dt * 1e3, "train_frames_per_s": 32 / dt, "infer_frames_per_s": 32 / di, "final_loss": float(loss)}))
