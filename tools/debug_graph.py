"""Replay-vs-direct check of the hipGraph mode: same inputs, several forward/backward calls, outputs and grads must repeat."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixelwiseregression_amd import PixelwiseRegression
from pixelwiseregression_amd.synthetic import make_batch
dev = "cuda:0"
torch.manual_seed(0)
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
m = PixelwiseRegression(14, stage=2, label_size=64, features=128, level=4, norm_method="instance").to(dev).train()
m.set_precision(prec)
b = make_batch(4, 14, S=128, seed=3, device=dev)
ref = None
for it in range(6):
    m.zero_grad()
    res = m(b["img"], b["label_img"], b["mask"])
    loss = sum(torch.mean(torch.sum((uvd - b["uvd"]) ** 2, dim=2)) for (_, _, uvd) in res)
    loss.backward()
    torch.cuda.synchronize()
    g = torch.cat([p.grad.flatten().float() for p in m.parameters()]).clone()
    outs = [t.detach().clone() for r in res for t in r]
    if ref is None:
        ref = (outs, g)
    else:
        eo = max((a - c).abs().max().item() for a, c in zip(outs, ref[0]))
        eg = (g - ref[1]).abs().max().item()
        # per-parameter gradient mismatch
        bad = []
        off = 0
        for name, p in m.named_parameters():
            n = p.numel()
            d = (g[off:off + n] - ref[1][off:off + n]).abs().max().item()
            if d > 1e-6 * max(1.0, ref[1][off:off + n].abs().max().item()):
                bad.append((name, d))
            off += n
        print("call %d: out err %.3e grad err %.3e (|g| %.3e) bad params %d %s" % (it, eo, eg, ref[1].abs().max().item(), len(bad), bad[:4]))
    del res, loss
