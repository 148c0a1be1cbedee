"""Host time to ISSUE one train step from an idle GPU against the time until the GPU has finished it (BASELINE C2, bf16, native harness):
   python tools/host_issue.py [--debug-lib] [reps = 40]
sync; t0; step(); t1 (all launches issued); sync; t2.  t1 - t0 close to t2 - t0 = the host's launch rate bounds the step."""
import os, sys, time, json, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--debug-lib" in sys.argv:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import dbglib  # noqa: F401
import torch
from pixelwiseregression_amd import PixelwiseRegression
from pixelwiseregression_amd.train import TrainStep
from pixelwiseregression_amd import synthetic

args = [a for a in sys.argv[1:] if not a.startswith("--")]
reps = int(args[0]) if args else 40
dev = "cuda:0"
torch.manual_seed(0)
model = PixelwiseRegression(14, stage=2, label_size=64, features=128, level=4, norm_method="instance", heatmap_method="softmax").to(dev)
model.set_precision("bf16")
model.train()
batch = synthetic.make_batch(32, 14, 128, device=dev)
step = TrainStep(model)


def one():
    return step(batch["img"], batch["label_img"], batch["mask"], batch["uvd"], batch.get("heatmaps"), batch.get("depthmaps"))


for _ in range(10):
    one()
torch.cuda.synchronize()
iss, tot = [], []
for _ in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); one(); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    iss.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
# free-running loop for comparison
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps):
    one()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(json.dumps({"issue_ms_from_idle_median": round(statistics.median(iss), 3), "issue_ms_min": round(min(iss), 3),
                  "total_ms_from_idle_median": round(statistics.median(tot), 3),
                  "free_running_issue_ms_per_step": round((t1 - t0) / reps * 1e3, 3), "free_running_ms_per_step": round((t2 - t0) / reps * 1e3, 3),
                  "env": {k: v for k, v in os.environ.items() if k.startswith("PWR_")}}))
