"""Isolated timing of the parameter-gradient (side-stream) layers and the stem's data gradients at the BASELINE C2 shapes, each at the
engine's split count (HIP events, interleaved rounds):   python tools/bench_side.py [which = all] [iters = 20] [--debug-lib]
`which`: comma list of s2 (stride-2 stem conv), last (heads' 128 -> J conv), stem (64 -> 128 and 32 -> 64 at 128 x 128), pw (the 1x1
convs of the 64 x 64 ResBlock and the stage input conv), dg (the stem's data gradients), s2dg (the stride-2 conv's data gradient with
the norm-backward sums, as the stem's backward runs it; the time includes the wrapper's two allocations + NaN fill)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--debug-lib" in sys.argv:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import dbglib  # noqa: F401
import torch
from pixelwiseregression_amd import kernels as K

dev = "cuda:0"
args = [a for a in sys.argv[1:] if not a.startswith("--")]
which = set((args[0] if args else "all").split(","))
iters = int(args[1]) if len(args) > 1 else 20
B = 32


def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def act(H, C):
    return torch.randn(B, H, H, C, device=dev).to(torch.bfloat16)


def wg(tag, H, Cin, Cout, k, stride, splits, norm=True, cout_real=None):
    x, dy = act(H, Cin), act(H // stride, Cout)
    st = K.norm_stats(x, torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev), mode=0) if norm else None
    ts = [timeit(lambda: K.conv_wgrad(x, dy, cout_real or Cout, k, stride, norm=st, splits=splits)) for _ in range(3)]
    fl = 2.0 * B * (H // stride) ** 2 * Cin * Cout * k * k
    print(json.dumps({"layer": tag, "splits": splits, "us": [round(t, 1) for t in ts], "TFLOPs": round(fl / min(ts) / 1e6, 1)}), flush=True)


def env_splits(name, dflt):
    return int(os.environ.get(name, dflt))


if which & {"all", "s2"}:
    for s in (env_splits("S2_SPLITS", 80), 40, 56):
        wg("stem stride-2 3x3 128->128, 128x128 -> 64x64 (wgrad + reduce)", 128, 128, 128, 3, 2, s)
if which & {"all", "last"}:
    wg("heads' last conv 3x3 128->16 (J = 14 padded) @64x64", 64, 128, 16, 3, 1, 80, cout_real=14)
if which & {"all", "stem"}:
    for sp in [int(v) for v in os.environ.get("STEM_SPLITS", "80").split(",")]:
        wg("stem 3x3 64->128 @128x128", 128, 64, 128, 3, 1, sp)
    for sp in [int(v) for v in os.environ.get("STEM_SPLITS2", "168").split(",")]:
        wg("stem 3x3 32->64 @128x128", 128, 32, 64, 3, 1, sp)
if which & {"all", "pw"}:
    wg("ResBlock 1x1 128->64 @64x64", 64, 128, 64, 1, 1, 512)
    wg("ResBlock 1x1 64->128 @64x64", 64, 64, 128, 1, 1, 512)
    wg("stage conv 1x1 128->128 @64x64 (no norm)", 64, 128, 128, 1, 1, 512, norm=False)
    wg("ResBlock 1x1 128->64 @32x32", 32, 128, 64, 1, 1, 128)
    wg("ResBlock 3x3 64->64 @64x64", 64, 64, 64, 3, 1, 80)
if which & {"all", "dg"}:
    for (tag, H, Cin, Cout) in (("stem data gradient 3x3 128->64 @128x128", 128, 128, 64), ("stem data gradient 3x3 64->32 @128x128", 128, 64, 32),
                                ("stem forward 3x3 64->128 @128x128", 128, 64, 128)):
        x = act(H, Cin)
        w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.03
        pack = K.pack_conv(w, 0, K.BF16)
        ts = [timeit(lambda: K.conv_fwd(x, pack, Cout, 3, 1)) for _ in range(3)]
        fl = 2.0 * B * H * H * Cin * Cout * 9
        print(json.dumps({"layer": tag, "us": [round(t, 1) for t in ts], "TFLOPs": round(fl / min(ts) / 1e6, 1)}), flush=True)
if which & {"all", "s2dg"}:
    dy, y = act(64, 128), act(128, 128)
    w = torch.randn(128, 128, 3, 3, device=dev) * 0.03
    pack = K.pack_conv(w, 2, K.BF16)
    st = K.norm_stats(y, torch.ones(128, device=dev), torch.zeros(128, device=dev), mode=0)
    ts = [timeit(lambda: K.conv_fwd_stats(dy, pack, 128, 3, 1, mode=1, nb_y=y, nb_state=st)) for _ in range(3)]
    fl = 2.0 * B * 64 * 64 * 128 * 128 * 9
    print(json.dumps({"layer": "stem stride-2 data gradient 128 <- 128, 64x64 -> 128x128 (+ norm-backward sums)", "us": [round(t, 1) for t in ts],
                      "TFLOPs": round(fl / min(ts) / 1e6, 1), "switches": {k: v for k, v in os.environ.items() if k.startswith("PWR_TR2")}}), flush=True)
