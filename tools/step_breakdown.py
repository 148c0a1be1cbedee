"""Where a train step's CHAIN time goes, by network part: forward {stem, stage-input, hourglass levels, heads, decoder} and every backward
segment {decoder, heads, hourglass levels, stage input, stem}, from events the DEBUG build of the engine puts at every change of scope in its
launch lists (pwr_engine_set_timing / pwr_engine_timing_report).  Train steps only -- no inference loop, no probes in the same process.
The side streams' weight-gradient kernels are not in these numbers (they run beside the chain); `step_ms` is the untimed step for scale.
    python tools/step_breakdown.py [steps=30]  ->  JSON (profiles/r6_step_breakdown.json)"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import dbglib  # noqa: F401
import torch
from pixelwiseregression_amd import PixelwiseRegression, _lib
from pixelwiseregression_amd.synthetic import make_batch
from pixelwiseregression_amd.train import TrainStep

N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = PixelwiseRegression(14, stage=2, label_size=64, features=128, level=4, norm_method="instance").to(dev).set_precision("bf16").train()
ts = TrainStep(m, opt="adam", lr=1e-4)
b = make_batch(32, 14, S=128, seed=1234, device=dev)
args = (b["img"], b["label_img"], b["mask"], b["uvd"])
for _ in range(10):
    ts(*args)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(N):
    ts(*args)
e1.record(); torch.cuda.synchronize()
step_ms = e0.elapsed_time(e1) / N
l = _lib.lib()
h = [p for p in m._engine.values() if p.need_grad][-1].h          # the training plan of this batch size (engine.py: _get_plan)
l.pwr_engine_set_timing.argtypes = [ctypes.c_void_p, ctypes.c_int]
l.pwr_engine_timing_report.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
l.pwr_engine_timing_report.restype = ctypes.c_size_t
l.pwr_engine_set_timing(h, 1)
e0.record()
for _ in range(N):
    ts(*args)
e1.record(); torch.cuda.synchronize()
timed_ms = e0.elapsed_time(e1) / N
buf = ctypes.create_string_buffer(1 << 16)
l.pwr_engine_timing_report(h, buf, len(buf))
l.pwr_engine_set_timing(h, 0)
out = {"what": "chain time per network part, ms per train step (BASELINE C2, B = 32, bf16, AdamW), mean of %d steps; events on the caller's stream at "
               "every change of scope; side-stream weight-gradient kernels run beside and are not included" % N,
       "step_ms": round(step_ms, 4), "step_ms_with_the_events": round(timed_ms, 4), "phases": {}}
for line in buf.value.decode().splitlines():
    phase, scope, ms, n = line.split("\t")
    out["phases"].setdefault(phase, {})[scope] = round(float(ms) / N, 4)


def group(d):
    g = {}
    for k, v in d.items():
        part = ("stem" if k.startswith("stem") else "decoder" if ".dec" in k else "stage input" if k.endswith(".in") else
                "hourglass" if ".hg" in k else "heads" if (".plane" in k or ".depth" in k or ".heads" in k) else k)
        g[part] = round(g.get(part, 0.0) + v, 4)
    return g


out["summary"] = {ph: dict(group(d), total=round(sum(d.values()), 4)) for ph, d in out["phases"].items()}
print(json.dumps(out, indent=1))
