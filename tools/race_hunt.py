"""Catch the rare non-reproducible train step IN THE ACT at full speed.

Like tools/determinism.py (N identical lr-0 steps, every step's flat gradient compared with step 0's on the device), plus: in the
first step whose gradient differs, the whole engine arena, the step's forward outputs and the loss gradients are snapshotted ON THE
DEVICE by pwr_debug_copy_if (an empty launch in every other step -- tools/determinism_arena.py compared 2 x 2.3 GB per step and
ran at 11.6 instead of 7.1 ms/step, which changes the timing the race depends on).  Afterwards every named arena buffer of the
snapshot is compared with a reference step's.

    PWR_JOIN_ONCE=1 python tools/race_hunt.py [steps]        # the configuration in which the events of round 1 were seen
"""
import os, sys, time, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dbglib  # noqa: F401  (the debug build: pwr_debug.h entry points, PWR_* experiment switches)
from pixelwiseregression_amd import PixelwiseRegression, _lib
from pixelwiseregression_amd.synthetic import make_batch
from pixelwiseregression_amd.train import TrainStep
dev = "cuda:0"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
B, J = 32, 14
torch.manual_seed(0)
m = PixelwiseRegression(J, stage=2, label_size=64, features=128, level=4, norm_method="instance").to(dev).set_precision("bf16").train()
b = make_batch(B, J, S=128, seed=1234, device=dev)
ts = TrainStep(m, opt="sgd", lr=0.0)
args = (b["img"], b["label_img"], b["mask"], b["uvd"])
ts(*args); ts(*args)
torch.cuda.synchronize()
plan = [p for p in m._engine.values() if p.need_grad][0]
l = _lib.lib()
l.pwr_engine_set_join(plan.h, 0 if os.environ.get("PWR_JOIN_ONCE", "0") not in ("", "0") else 1)   # (debug build only: pwr_debug.h)
need = l.pwr_engine_layout(plan.h, None, 0)
buf = ctypes.create_string_buffer(need)
l.pwr_engine_layout(plan.h, buf, need)
recs = [(int(o), int(n), tag) for o, n, tag in (line.split(" ", 2) for line in buf.value.decode().splitlines())]
arena = plan.arena
nb = arena.numel() // 16 * 16
a0 = arena[:nb].clone()
g0 = m.flat_grad().clone()
outs0 = [t.clone() for t in ts._keep[0]]
gu0 = [ts._scratch[1]["gu%d" % s].clone() for s in range(2)]
cap = torch.zeros_like(a0)
cap_outs = [torch.zeros_like(t) for t in outs0]
cap_gu = [torch.zeros_like(t) for t in gu0]
cap_g = torch.zeros_like(g0)
have = torch.zeros((), dtype=torch.bool, device=dev)
flag = torch.zeros(1, dtype=torch.int32, device=dev)
gbad = torch.zeros(N, device=dev)
stream = _lib.stream_ptr(torch.device(dev))


def copy_if(src, dst):
    nbytes = src.numel() * src.element_size()
    assert nbytes % 16 == 0, nbytes
    _lib.check(l.pwr_debug_copy_if(flag.data_ptr(), src.data_ptr(), dst.data_ptr(), nbytes, stream), "copy_if")


torch.cuda.synchronize()
t0 = time.time()
for it in range(N):
    ts(*args)
    fb = (m.flat_grad() != g0).any()
    gbad[it] = fb
    flag.copy_((fb & ~have).to(torch.int32).view(1))
    have |= fb
    copy_if(arena[:nb], cap)
    copy_if(m.flat_grad()[:g0.numel() // 4 * 4], cap_g[:g0.numel() // 4 * 4])
    for t, c in zip(ts._keep[0], cap_outs):
        copy_if(t, c)
    for s in range(2):
        copy_if(ts._scratch[1]["gu%d" % s], cap_gu[s])
torch.cuda.synchronize()
bad = gbad.nonzero().flatten().tolist()
print("join_once=%s  steps %d, %.2f ms/step; steps with a different gradient: %d %s" % (os.environ.get("PWR_JOIN_ONCE", "0"), N, (time.time() - t0) / N * 1e3,
                                                                                       len(bad), bad[:20]))
if not bad:
    sys.exit(0)
print("first bad step: %d -- snapshot of that step vs a reference step" % bad[0])
names = ["s0.heatmaps", "s0.depthmaps", "s0.uvd", "s1.heatmaps", "s1.depthmaps", "s1.uvd"]
for nm, x, y in list(zip(names, cap_outs, outs0)) + [("gU.s0", cap_gu[0], gu0[0]), ("gU.s1", cap_gu[1], gu0[1])]:
    d = (x != y)
    if bool(d.any()):
        idx = d.nonzero()
        print("  forward output / loss gradient %-14s differs in %d elements, first at %s: %.8g vs %.8g" % (nm, idx.shape[0], idx[0].tolist(), float(x[tuple(idx[0])]),
                                                                                                     float(y[tuple(idx[0])])))
capb, a0b = cap.view(torch.uint8), a0.view(torch.uint8)
dump = {"bad_steps": bad, "maps": []}
tags = {t.split(":")[0] + ":" + t.split(":")[1].split("#")[0]: (o, n) for o, n, t in recs}
def arena_f32(buf, key):
    o, n = tags[key]
    return buf[o:o + n].view(torch.float32).view(B, J, -1)
for st in (0, 1):
    k = "s%d.dec:gz" % st
    if k not in tags:
        continue
    dd = (arena_f32(capb, k) != arena_f32(a0b, k)).any(2).nonzero().tolist()
    for (bb, jj) in dd[:4]:
        rec = {"stage": st, "b": bb, "j": jj, "w": m.stages[st].plane_regression.w.detach().cpu().clone(),
               "label_img": b["label_img"][bb].cpu(), "mask": b["mask"][bb].cpu()}
        for nm in ("z", "gz", "gDt", "gH", "gD"):
            kk = "s%d.dec:%s" % (st, nm)
            if kk in tags:
                rec["cap_" + nm], rec["ref_" + nm] = arena_f32(capb, kk)[bb, jj].cpu().clone(), arena_f32(a0b, kk)[bb, jj].cpu().clone()
        for nm, x, y in zip(names, cap_outs, outs0):
            if nm.startswith("s%d." % st):
                rec["cap_" + nm[3:]], rec["ref_" + nm[3:]] = x[bb, jj].cpu().clone(), y[bb, jj].cpu().clone()
        rec["cap_gU"], rec["ref_gU"] = cap_gu[st][bb, jj].cpu().clone(), gu0[st][bb, jj].cpu().clone()
        dump["maps"].append(rec)
out_path = os.environ.get("RACE_DUMP", "")
if out_path:
    torch.save(dump, out_path)
    print("raw decoder buffers of the captured step saved to", out_path)
for o, n, tag in recs:
    if o + n > nb:
        continue
    x, y = capb[o:o + n], a0b[o:o + n]
    if n == 0 or bool((x == y).all()):
        continue
    name = tag.split(":")[1].split("#")[0]
    line = "  %-44s %9d bytes differ" % (tag, int((x != y).sum()))
    if name in ("z", "gz", "gDt", "gH", "gD") and n == B * J * 4096 * 4:
        xf, yf = x.view(torch.float32).view(B, J, -1), y.view(torch.float32).view(B, J, -1)
        dd = (xf != yf).nonzero()
        line += "  elements %d; samples %s joints %s pixels %s" % (dd.shape[0], dd[:, 0].unique().tolist()[:6], dd[:, 1].unique().tolist()[:10], dd[:, 2].unique().tolist()[:12])
        j = tuple(dd[0].tolist()); line += "  e.g. %.8g vs %.8g" % (float(xf[j]), float(yf[j]))
    elif name == "gwp":
        xf, yf = x.view(torch.float32), y.view(torch.float32)
        line += "  entries (b*J+j) %s" % (xf != yf).nonzero().flatten().tolist()[:8]
    elif (name.startswith("grd") or name.startswith("act")) and "x" in name:
        C = int(name.split("x")[-1])
        HW = n // (B * C * 2)
        if HW * B * C * 2 == n:
            xf, yf = x.view(torch.bfloat16).view(B, HW, C), y.view(torch.bfloat16).view(B, HW, C)
            dd = (xf != yf).nonzero()
            line += "  elements %d; samples %s channels %s pixels %s" % (dd.shape[0], dd[:, 0].unique().tolist()[:6], dd[:, 2].unique().tolist()[:10], dd[:, 1].unique().tolist()[:12])
            j = tuple(dd[0].tolist()); line += "  e.g. %.6g vs %.6g" % (float(xf[j]), float(yf[j]))
    print(line)
