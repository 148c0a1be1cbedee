"""Diagnosis: gradient of the trained fixture's linear functional -- fp32 engine, bf16 engine (old / new backward paths via the debug
build's switches), stock bf16 autocast -- against the reference's float64 gradient, overall and by parameter group."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dbglib  # noqa: F401
import numpy as np, torch
from pixelwiseregression_amd import PixelwiseRegression
DEV = "cuda:0"
g = np.load(os.path.join(ROOT, "tests", "golden", "trained_c2.npz"))
m = PixelwiseRegression(14, stage=2, label_size=64, features=128, level=4, kernel_size=3, norm_method="instance", heatmap_method="softmax")
m.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd_")}, strict=True)
m = m.to(DEV).train()
b = {k[3:]: torch.from_numpy(g[k]).to(DEV) for k in g.files if k.startswith("in_")}
ref = torch.from_numpy((g["f64_grad_bf16bits"].astype(np.uint32) << 16).view(np.float32)).double()
keys, numel = [str(k) for k in g["grad_keys"]], [int(n) for n in g["grad_numel"]]
up = lambda a: torch.from_numpy(np.kron(a, np.ones((8, 8), dtype=np.float32))).to(DEV)
def functional(res):
    return sum((u_ * torch.from_numpy(g["GU%d" % s_]).to(DEV)).sum() + (p_ * up(g["GH%d" % s_])).sum() + (D_ * up(g["GD%d" % s_])).sum()
               for s_, (p_, D_, u_) in enumerate(res))
def grad_of(fwd):
    m.zero_grad(set_to_none=True)
    functional(fwd(b["img"], b["label_img"], b["mask"])).backward()
    return torch.cat([p.grad.detach().flatten() for _, p in m.named_parameters()]).double().cpu()
def report(name, got):
    rel = float((got - ref).norm() / ref.norm()); cos = float(torch.dot(got, ref) / (got.norm() * ref.norm()))
    groups = {}
    o = 0
    for k, n in zip(keys, numel):
        grp = k.split(".")[0] if k.startswith("conv") else ".".join(k.split(".")[:3])[:40]
        d = groups.setdefault(grp, [0.0, 0.0])
        d[0] += float((got[o:o + n] - ref[o:o + n]).pow(2).sum()); d[1] += float(ref[o:o + n].pow(2).sum())
        o += n
    print("%-34s rel %.3e cos %.5f | %s" % (name, rel, cos, "  ".join("%s %.2e/%.2e" % (k, v[0] ** 0.5, v[1] ** 0.5) for k, v in groups.items())), flush=True)
from aten_reference import aten_forward
m.set_precision("fp32"); report("fp32 engine", grad_of(m))
m.set_precision("bf16")
os.environ["PWR_WGRAD3W"] = "0"; os.environ["PWR_HEAD_BWD_PAIR"] = "0"
report("bf16 engine, round-3 paths", grad_of(m))
report("stock bf16 autocast", grad_of(lambda *a: aten_forward(m, *a)))
m2 = PixelwiseRegression(14, stage=2, label_size=64, features=128, level=4, kernel_size=3, norm_method="instance", heatmap_method="softmax")
m2.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd_")}, strict=True)
os.environ["PWR_WGRAD3W"] = "1"; os.environ["PWR_HEAD_BWD_PAIR"] = "1"
m = m2.to(DEV).train().set_precision("bf16")          # (a new module: a new plan, built with the switches above)
report("bf16 engine, round-4 paths", grad_of(m))
