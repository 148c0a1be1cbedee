"""Debug: replay one ResBlock's backward kernels (dgrad 3x3, norm_bwd) on the real data of a failing config."""
import sys, os
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import model_ref
from weights_util import fill_state_dict
from pixelwiseregression_amd import PixelwiseRegression, kernels as K
from pixelwiseregression_amd.synthetic import make_batch

TARGET = sys.argv[1] if len(sys.argv) > 1 else "stages.1.hourglass.inner.output_conv"
stash = {}
orig = model_ref._resblock

def patched(x, sd, prefix, cfg, tr, bu):
    if prefix != TARGET:
        return orig(x, sd, prefix, cfg, tr, bu)
    n0 = model_ref._norm(x, sd, prefix + ".conv.0", cfg, tr, bu); r0 = F.relu(n0)
    t1 = model_ref._conv(r0, sd, prefix + ".conv.2")
    n1 = model_ref._norm(t1, sd, prefix + ".conv.3", cfg, tr, bu); r1 = F.relu(n1)
    t2 = model_ref._conv(r1, sd, prefix + ".conv.5", padding=1)
    n2 = model_ref._norm(t2, sd, prefix + ".conv.6", cfg, tr, bu); r2 = F.relu(n2)
    out = model_ref._conv(r2, sd, prefix + ".conv.8")
    for k, v in dict(x=x, t1=t1, r1=r1, t2=t2, r2=r2, out=out).items():
        v.retain_grad(); stash[k] = v
    return x + out

model_ref._resblock = patched
cfg = dict(J=4, B=3, P=16, F=32, level=2, norm="instance")
m = PixelwiseRegression(4, stage=2, label_size=16, features=32, level=2, norm_method="instance")
sd = fill_state_dict(m.state_dict(), seed=7)
batch = make_batch(3, 4, S=32, seed=4)
rc = model_ref.RefConfig(4, 2, 16, 32, 2, 3, "instance", "softmax")
params = {k: (v.double() if v.is_floating_point() else v).clone() for k, v in sd.items()}
for k, v in params.items():
    if v.is_floating_point() and "running" not in k and "filter" not in k:
        v.requires_grad_()
res = model_ref.forward(params, rc, batch["img"].double(), batch["label_img"].double(), batch["mask"].double(), training=True, bn_updates={})
model_ref.train_loss(res, batch["uvd"].double()).backward()
DEV = "cuda:0"
nhwc = lambda t: t.detach().permute(0, 2, 3, 1).contiguous().float().to(DEV)
nchw = lambda t: t.double().cpu().permute(0, 3, 1, 2)
t1, r1, t2 = stash["t1"], stash["r1"], stash["t2"]
print("shapes", tuple(t1.shape), "t2.grad max", float(t2.grad.abs().max()), "r1.grad max", float(r1.grad.abs().max()))
# (1) dgrad of conv.5: input t2.grad -> expected r1.grad
w5 = params[TARGET + ".conv.5.weight"].detach()
pack = K.pack_conv(w5.float().to(DEV), 1, K.F32)
g, _ = K.conv_fwd(nhwc(t2.grad), pack, w5.shape[1], 3, 1)
print("dgrad err", float((nchw(g) - r1.grad).abs().max()), "ref max", float(r1.grad.abs().max()))
# (2) norm bwd on t1: g = r1.grad -> expected t1.grad ; dgamma/dbeta
gam, bet = params[TARGET + ".conv.3.weight"], params[TARGET + ".conv.3.bias"]
st = K.norm_stats(nhwc(t1), gam.detach().float().to(DEV), bet.detach().float().to(DEV), mode=0)
dy, dg, db = K.norm_bwd(nhwc(r1.grad), nhwc(t1), st, relu=True)
print("norm_bwd dy err", float((nchw(dy) - t1.grad).abs().max()), "ref max", float(t1.grad.abs().max()))
print("dgamma err", float((dg.double().cpu() - gam.grad).abs().max()), "max", float(gam.grad.abs().max()))
print("dbeta err", float((db.double().cpu() - bet.grad).abs().max()), "max", float(bet.grad.abs().max()))
n1 = F.instance_norm(t1.detach(), None, None, gam.detach(), bet.detach(), True, 0.1, 1e-5)
print("min |norm(t1)|", float(n1.abs().min()), "count <1e-4:", int((n1.abs() < 1e-4).sum()))
v = t1.detach().var(dim=(2, 3), unbiased=False)
print("min var", float(v.min()), "max rstd", float((v.min() + 1e-5).rsqrt()))
