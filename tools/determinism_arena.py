"""Like determinism.py, but compares the engine's whole ARENA (every activation, gradient, scratch and -- with PWR_DEBUG_NB=1 -- the
debug copies of the norm-backward partial sums of the heads) with step 0's after every step, on the device; the first differing
arena is kept and analysed buffer by buffer (pwr_engine_layout names them).

    PWR_DEBUG_NB=1 python tools/determinism_arena.py [steps]
"""
import os, sys, time, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dbglib  # noqa: F401  (the debug build: pwr_debug.h entry points, PWR_* experiment switches)
from pixelwiseregression_amd import PixelwiseRegression, _lib
from pixelwiseregression_amd.synthetic import make_batch
from pixelwiseregression_amd.train import TrainStep
dev = "cuda:0"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
torch.manual_seed(0)
B = 32
m = PixelwiseRegression(14, stage=2, label_size=64, features=128, level=4, norm_method="instance").to(dev).set_precision("bf16").train()
b = make_batch(B, 14, S=128, seed=1234, device=dev)
ts = TrainStep(m, opt="sgd", lr=0.0)
args = (b["img"], b["label_img"], b["mask"], b["uvd"])
ts(*args); ts(*args)
torch.cuda.synchronize()
plan = [p for p in m._engine.values() if p.need_grad][0]
l = _lib.lib()
need = l.pwr_engine_layout(plan.h, None, 0)
buf = ctypes.create_string_buffer(need)
l.pwr_engine_layout(plan.h, buf, need)
recs = []
for line in buf.value.decode().splitlines():
    o, n, tag = line.split(" ", 2)
    recs.append((int(o), int(n), tag))
arena = plan.arena
nblk = arena.numel() // 256
a64 = arena[:nblk * 256].view(torch.int64).view(nblk, 32)
rid = torch.full((nblk,), len(recs), dtype=torch.long, device=dev)
for i, (o, n, tag) in enumerate(recs):
    rid[o // 256:(o + n + 255) // 256] = i
a0 = a64.clone()
g0 = m.flat_grad().clone()
cap = torch.zeros_like(a0)
have = torch.zeros((), dtype=torch.bool, device=dev)
counts = torch.zeros(N, len(recs) + 1, device=dev)
gbad = torch.zeros(N, device=dev)
torch.cuda.synchronize()
t0 = time.time()
for it in range(N):
    ts(*args)
    d = (a64 != a0).any(1)
    counts[it].index_add_(0, rid, d.float())
    fb = (m.flat_grad() != g0).any()
    gbad[it] = fb
    take = fb & ~have
    cap = torch.where(take, a64, cap)
    have |= take
torch.cuda.synchronize()
print("steps %d, %.2f ms/step; steps with a different gradient: %d; steps with a different arena: %d"
      % (N, (time.time() - t0) / N * 1e3, int(gbad.sum()), int((counts.sum(1) > 0).sum())))
always = (counts > 0).float().mean(0) > 0.5        # buffers that differ from step 0 most of the time (uninitialised padding etc.)
print("buffers that differ from step 0 in most steps:", [recs[i][2] for i in always.nonzero().flatten().tolist() if i < len(recs)][:20])
if bool(have):
    capb = cap.view(-1).view(torch.uint8)
    a0b = a0.view(-1).view(torch.uint8)
    it = int(gbad.nonzero()[0])
    print("first bad step:", it)
    for i in counts[it].nonzero().flatten().tolist():
        if i >= len(recs) or bool(always[i]): continue
        o, n, tag = recs[i]
        x, y = capb[o:o + n], a0b[o:o + n]
        line = "  %-40s blocks %6d" % (tag, int(counts[it, i]))
        name = tag.split(":")[1].split("#")[0]
        if name in ("dbg_partial",):
            xf, yf = x.view(torch.float32).view(B, -1, 2, 128), y.view(torch.float32).view(B, -1, 2, 128)
            dd = (xf != yf).nonzero()
            line += "  entries (b, chunk, which, c): %d %s" % (dd.shape[0], dd[:6].tolist())
            if dd.shape[0]:
                j = tuple(dd[0].tolist()); line += "  e.g. %.6g vs %.6g" % (float(xf[j]), float(yf[j]))
        elif name in ("dbg_S",):
            xf, yf = x.view(torch.float32).view(2, B, 128), y.view(torch.float32).view(2, B, 128)
            dd = (xf != yf).nonzero()
            line += "  entries (which, b, c): %d %s" % (dd.shape[0], dd[:6].tolist())
            if dd.shape[0]:
                j = tuple(dd[0].tolist()); line += "  e.g. %.6g vs %.6g" % (float(xf[j]), float(yf[j]))
        elif name.startswith("grd") or name.startswith("act") or name == "dbg_g":
            C = 128 if name == "dbg_g" else int(name.split("x")[-1])
            HW = n // (B * C * 2)
            xf, yf = x.view(torch.bfloat16).view(B, HW, C), y.view(torch.bfloat16).view(B, HW, C)
            dd = (xf != yf).nonzero()
            line += "  elements %d; samples %s channels %s pixels %s" % (dd.shape[0], dd[:, 0].unique().tolist()[:6], dd[:, 2].unique().tolist()[:10],
                                                                        dd[:, 1].unique().tolist()[:10])
            if dd.shape[0]:
                j = tuple(dd[0].tolist()); line += "  e.g. %.6g vs %.6g" % (float(xf[j]), float(yf[j]))
        elif name in ("z", "gz", "gDt", "gH", "gD"):
            xf, yf = x.view(torch.float32).view(B, 14, -1), y.view(torch.float32).view(B, 14, -1)
            dd = (xf != yf).nonzero()
            line += "  elements %d; samples %s joints %s pixels %s" % (dd.shape[0], dd[:, 0].unique().tolist()[:6], dd[:, 1].unique().tolist()[:10],
                                                                      dd[:, 2].unique().tolist()[:10])
            if dd.shape[0]:
                j = tuple(dd[0].tolist()); line += "  e.g. %.6g vs %.6g" % (float(xf[j]), float(yf[j]))
        elif name in ("gwp",):
            xf, yf = x.view(torch.float32), y.view(torch.float32)
            dd = (xf != yf).nonzero().flatten()
            line += "  entries %s" % dd.tolist()[:8]
        print(line)
