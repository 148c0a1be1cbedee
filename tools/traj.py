"""Per-step loss trajectory of the bench's training loop (AdamW, lr 1e-3 like bench.py) -> file; compare trajectories of several processes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "run":
    from pixelwiseregression_amd import PixelwiseRegression
    from pixelwiseregression_amd.synthetic import make_batch
    from pixelwiseregression_amd.train import TrainStep
    dev = "cuda:0"
    torch.manual_seed(0)
    m = PixelwiseRegression(14, stage=2, label_size=64, features=128, level=4, norm_method="instance").to(dev).set_precision("bf16").train()
    b = make_batch(32, 14, S=128, seed=1234, device=dev)
    ts = TrainStep(m, opt="adam", lr=1e-3)
    N = int(sys.argv[3])
    losses = torch.zeros(N, device=dev); gsum = torch.zeros(N, device=dev, dtype=torch.float64)
    for it in range(N):
        ts(b["img"], b["label_img"], b["mask"], b["uvd"])
        losses[it] = ts.loss[0]; gsum[it] = m.flat_grad().double().abs().sum()
    torch.save({"loss": losses.cpu(), "gsum": gsum.cpu()}, sys.argv[2])
else:
    ref = torch.load(sys.argv[2])
    for f in sys.argv[3:]:
        o = torch.load(f)
        d = ((o["loss"] != ref["loss"]) | (o["gsum"] != ref["gsum"])).nonzero().flatten()
        print(f, "identical" if d.numel() == 0 else "first difference at step %d (loss %.6f vs %.6f, |g| %.6e vs %.6e)" % (int(d[0]), o["loss"][d[0]], ref["loss"][d[0]], o["gsum"][d[0]], ref["gsum"][d[0]]))
