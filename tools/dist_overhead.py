"""Where does the one-rank data-parallel path lose time against the plain path?  (round-2 review: 7.90 vs 6.9 ms/step, +14 %, with no
communication at all.)  Same process, same model: the plain TrainStep, then the same with ddp.DataParallel attached (process group
over RCCL, world 1), each timed over `steps` steps -- wall time until the GPU has finished and the time the HOST needs to issue the
steps -- plus variants of the hook (see ddp.py).

    python tools/dist_overhead.py [steps] > profiles/r3_dist_overhead.json
"""
import json, os, sys, time
import torch
import torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pixelwiseregression_amd import PixelwiseRegression
from pixelwiseregression_amd.synthetic import make_batch
from pixelwiseregression_amd.train import TrainStep
from pixelwiseregression_amd import ddp as D

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
torch.manual_seed(0)
m = PixelwiseRegression(14, stage=2, label_size=64, features=128, level=4, norm_method="instance").to(dev).set_precision("bf16").train()
tr = TrainStep(m, opt="adam", lr=1e-4, beta1=0.9, beta2=0.999, weight_decay=0.0, alpha=1.0, lambda_h=1.0, lambda_d=0.01)
b = make_batch(32, 14, S=128, seed=1234, device=dev, dense_targets=True)
step = lambda: tr(b["img"], b["label_img"], b["mask"], b["uvd"], b["heatmaps"], b["depthmaps"])


def timed(tag):
    for _ in range(30):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    r = {"variant": tag, "ms_per_step": t_all / steps * 1e3, "host_issue_ms_per_step": t_issue / steps * 1e3}
    print(json.dumps(r), file=sys.stderr, flush=True)
    return r


out = {"what": "train step BASELINE configs[1], one rank; %d timed steps after 30 warm-up per variant, same process" % steps, "variants": []}
out["variants"].append(timed("single (no process-group hook)"))
for mode in D.MODES:
    D.DataParallel(m, broadcast=False, mode=mode)
    out["variants"].append(timed("dp1, hook mode '%s'" % mode))
    m._ddp = None
out["variants"].append(timed("single again (drift)"))
base = out["variants"][0]["ms_per_step"]
for v in out["variants"]:
    v["vs_single"] = v["ms_per_step"] / base
print(json.dumps(out, indent=1))
dist.destroy_process_group()
