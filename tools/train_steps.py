"""N train steps of the bench configuration (BASELINE C2, B = 32, bf16, AdamW) and nothing else -- no inference loop, no roofline probe, no
vendor-GEMM yardstick in the same process: the program rocprofv3 wraps for the per-step kernel statistics (calls / N = launches per step).
    python tools/train_steps.py [N=40] [--debug-lib]      (--debug-lib: the debug build, whose PWR_* switches are live)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--debug-lib" in sys.argv:
    sys.argv.remove("--debug-lib")
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import dbglib  # noqa: F401
import torch
from pixelwiseregression_amd import PixelwiseRegression
from pixelwiseregression_amd.synthetic import make_batch
from pixelwiseregression_amd.train import TrainStep
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = PixelwiseRegression(14, stage=2, label_size=64, features=128, level=4, norm_method="instance").to(dev).set_precision("bf16").train()
ts = TrainStep(m, opt="adam", lr=1e-4)
b = make_batch(32, 14, S=128, seed=1234, device=dev)
a = (b["img"], b["label_img"], b["mask"], b["uvd"])
ts(*a)                                   # (plan creation, first pack)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(N):
    ts(*a)
e1.record(); torch.cuda.synchronize()
print(json.dumps({"train_steps": N + 1, "ms_per_step_in_this_process": e0.elapsed_time(e1) / N}))
