"""Same weights, same batch: gradients / outputs of a few AdamW steps under different env settings must agree bitwise.
    python tools/cmp_modes.py dump out.pt     (run under the env to test)
    python tools/cmp_modes.py cmp a.pt b.pt"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "dump":
    from pixelwiseregression_amd import PixelwiseRegression
    from pixelwiseregression_amd.synthetic import make_batch
    from pixelwiseregression_amd.train import TrainStep
    dev = "cuda:0"
    torch.manual_seed(0)
    m = PixelwiseRegression(14, stage=2, label_size=64, features=128, level=4, norm_method="instance").to(dev).set_precision("bf16").train()
    b = make_batch(32, 14, S=128, seed=1234, device=dev)
    ts = TrainStep(m, opt="adam", lr=1e-3)
    out = {}
    for it in range(4):
        ts(b["img"], b["label_img"], b["mask"], b["uvd"])
        out["g%d" % it] = m.flat_grad().clone().cpu()
        out["l%d" % it] = ts.loss.clone().cpu()
    m.eval()
    with torch.no_grad():
        r = m(b["img"], b["label_img"], b["mask"])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): r = m(b["img"], b["label_img"], b["mask"])
        torch.cuda.synchronize(); out["infer_ms"] = (time.perf_counter() - t0) / 50 * 1e3
    out["uvd"] = r[-1][2].cpu(); out["p"] = r[-1][0].cpu()
    torch.save(out, sys.argv[2])
    print("inference %.3f ms" % out["infer_ms"])
else:
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in a:
        if k == "infer_ms": continue
        d = (a[k] != b[k]).sum().item()
        print(k, "different elements:", d, "" if d == 0 else "max |diff| %.3e" % (a[k] - b[k]).abs().max().item())
    if len(sys.argv) > 4:   # per-parameter report for g0
        from pixelwiseregression_amd import PixelwiseRegression
        m = PixelwiseRegression(14, stage=2, label_size=64, features=128, level=4, norm_method="instance")
        bad = a["g0"] != b["g0"]
        for name, (o, shape) in m._offsets.items():
            n = int(torch.Size(shape).numel()); c = int(bad[o:o + n].sum())
            if name.startswith("stages.1") and ("regression" in name): print("   ", name, c, "/", n)
