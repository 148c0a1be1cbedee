"""Forward accuracy of the fp32 engine against the float64 evaluation of the reference, next to the reference's own fp32 run
(tests/golden/wellcond.npz): max |x - x_f64| per output, for the engine and for the reference."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from weights_util import fill_state_dict
from pixelwiseregression_amd import PixelwiseRegression
g = np.load(os.path.join(ROOT, "tests", "golden", "wellcond.npz"))
for tag in "abc":
    pre = tag + "_"
    kw = {k: (str(g[pre + "cfg_" + k]) if k.endswith("method") else int(g[pre + "cfg_" + k]))
          for k in ("stage", "label_size", "features", "level", "kernel_size", "norm_method", "heatmap_method")}
    m = PixelwiseRegression(int(g[pre + "cfg_joints"]), **kw)
    m.load_state_dict(fill_state_dict(m.state_dict(), seed=int(g[pre + "weights_seed"])))
    m = m.to("cuda:0").set_precision("fp32").eval()
    b = {k[len(pre) + 3:]: torch.from_numpy(g[k]).to("cuda:0") for k in g.files if k.startswith(pre + "in_")}
    with torch.no_grad():
        m.train()
        res = m(b["img"], b["label_img"], b["mask"])
    for s, (p, D, uvd) in enumerate(res):
        for nm, t in (("p", p), ("D", D), ("uvd", uvd)):
            f64, f32 = g[pre + "f64_s%d_%s" % (s, nm)], g[pre + "f32_s%d_%s" % (s, nm)]
            e = np.abs(t.double().cpu().numpy() - f64).max(); r = np.abs(f32.astype(np.float64) - f64).max()
            print("fixture %s stage %d %-3s  engine %.2e  reference fp32 %.2e  ratio %5.1f   (max |value| %.2e)" % (tag, s, nm, e, r, e / max(r, 1e-30), np.abs(f64).max()))
