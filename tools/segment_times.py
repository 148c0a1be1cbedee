"""Where a train step's time goes, by phase: forward, loss, each backward segment (stage S-1 .. stage 0, stem; side streams joined at
the end of each), optimizer -- HIP events on the caller's stream around the C ABI calls (TrainStep's own sequence), averaged over steps.
    python tools/segment_times.py [--debug-lib]"""
import os, sys, json, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "--debug-lib" in sys.argv:
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import dbglib  # noqa: F401
import torch
from pixelwiseregression_amd import PixelwiseRegression, _lib
from pixelwiseregression_amd.synthetic import make_batch
from pixelwiseregression_amd.train import TrainStep
from pixelwiseregression_amd import engine as E
dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = PixelwiseRegression(14, stage=2, label_size=64, features=128, level=4, norm_method="instance").to(dev).set_precision("bf16").train()
ts = TrainStep(m, opt="adam", lr=1e-4)
b = make_batch(32, 14, S=128, seed=1234, device=dev)
marks = []
l = _lib.lib()
orig_bwd, orig_fwd, orig_adam = l.pwr_engine_backward, l.pwr_engine_forward, l.pwr_adamw_step
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
class Wrap:
    def __init__(self, f, tag): self.f, self.tag = f, tag
    def __call__(self, *a):
        marks.append((self.tag + ":begin", ev())); r = self.f(*a); marks.append((self.tag + ":end", ev())); return r
l.pwr_engine_forward = Wrap(orig_fwd, "forward")
l.pwr_engine_backward = Wrap(orig_bwd, "bwd")
l.pwr_adamw_step = Wrap(orig_adam, "adamw")
for _ in range(10):
    ts(b["img"], b["label_img"], b["mask"], b["uvd"])
torch.cuda.synchronize()
acc = {}
N = 50
for _ in range(N):
    marks.clear()
    t0 = ev()
    ts(b["img"], b["label_img"], b["mask"], b["uvd"])
    t1 = ev()
    torch.cuda.synchronize()
    names, k = [], 0
    seq = [("step:begin", t0)] + marks + [("step:end", t1)]
    seg = 0
    for (na, ea), (nb, eb) in zip(seq[:-1], seq[1:]):
        if na == "bwd:begin": key = "backward segment %d" % seg; seg += 1
        elif na == "forward:begin": key = "forward"
        elif na == "adamw:begin": key = "adamw"
        else: key = "between (%s -> %s)" % (na.split(":")[0], nb.split(":")[0])
        acc[key] = acc.get(key, 0.0) + ea.elapsed_time(eb)
    acc["whole step"] = acc.get("whole step", 0.0) + t0.elapsed_time(t1)
print(json.dumps({k: round(v / N, 4) for k, v in acc.items()}))
