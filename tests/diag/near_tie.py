"""Diagnostic (CPU, float64 oracle): which ReLU decisions of the well-conditioned fixtures sit within fp32 rounding of zero, and what
would ONE flipped decision do to the gradient of that norm's bias?  For every norm -> ReLU pair the pre-activation y and the
gradient g arriving at the ReLU's OUTPUT are recorded; flipping the decision at element e changes dbeta[c] by exactly g[e], so
|g[e]| / max|dbeta| is the error a single flip shows as in tools/grad_report.py's "worst tensors" column.

    python tests/diag/near_tie.py            # fixtures a, b, c of tests/golden/wellcond.npz
"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from weights_util import fill_state_dict
from oracle import model_ref
from pixelwiseregression_amd import PixelwiseRegression


def main():
    g = np.load(os.path.join(ROOT, "tests", "golden", "wellcond.npz"))
    for tag in "abc":
        pre = tag + "_"
        kw = {k: (str(g[pre + "cfg_" + k]) if k.endswith("method") else int(g[pre + "cfg_" + k]))
              for k in ("stage", "label_size", "features", "level", "kernel_size", "norm_method", "heatmap_method")}
        J = int(g[pre + "cfg_joints"])
        m = PixelwiseRegression(J, **kw)
        sd = fill_state_dict(m.state_dict(), seed=int(g[pre + "weights_seed"]))
        params = {k: (v.double() if v.is_floating_point() else v).clone() for k, v in sd.items()}
        for k, v in params.items():
            if v.is_floating_point() and "running" not in k and "filter" not in k:
                v.requires_grad_()
        b = {k[len(pre) + 3:]: torch.from_numpy(g[k]).double() for k in g.files if k.startswith(pre + "in_")}
        rec, last = [], {}
        norm0, relu0 = model_ref._norm, model_ref.F.relu

        def norm(x, sd_, prefix, *a):
            y = norm0(x, sd_, prefix, *a)
            last["y"], last["prefix"] = y, prefix
            return y

        def relu(x, *a, **k):
            r = relu0(x, *a, **k)
            if last.get("y") is x:
                r.retain_grad()
                rec.append((last["prefix"], x.detach(), r))
            return r

        model_ref._norm, model_ref.F.relu = norm, relu
        try:
            rc = model_ref.RefConfig(J, kw["stage"], kw["label_size"], kw["features"], kw["level"], kw["kernel_size"],
                                     kw["norm_method"], kw["heatmap_method"])
            res = model_ref.forward(params, rc, b["img"], b["label_img"], b["mask"], training=True, bn_updates={})
            model_ref.train_loss(res, b["uvd"], b["heatmaps"], b["depthmaps"], alpha=float(g[pre + "alpha"])).backward()
        finally:
            model_ref._norm, model_ref.F.relu = norm0, relu0
        rows = []
        for prefix, y, r in rec:
            dbeta = params[prefix + ".bias"].grad
            scale = float(y.abs().max())
            near = (y.abs() < 4e-6 * scale)          # a few fp32 ulps of the tensor's range
            if near.any():
                idx = near.nonzero()
                for i in idx:
                    e = tuple(int(v) for v in i)
                    rows.append((float(r.grad[e].abs() / dbeta.abs().max()), prefix, e, float(y[e]) / scale))
        rows.sort(reverse=True)
        print("fixture %s: %d norm->ReLU pairs, %d decisions within 4e-6 of the tensor range; largest single-flip effects on dbeta:"
              % (tag, len(rec), len(rows)))
        want = os.environ.get("NEAR_TIE_PREFIX")
        for rel, prefix, e, yy in ([r for r in rows if want in r[1]] if want else rows[:6]):
            print("   %-46s elem %-18s y/range %+.2e   |g|/max|dbeta| %.2e" % (prefix, e, yy, rel))


if __name__ == "__main__":
    main()
