"""Oracle (test infrastructure): functional CPU restatement of PixelwiseRegression.forward.

A state_dict-driven, plain-torch (ATen CPU: oneDNN/MKL) evaluation of the same op sequence
as /root/reference/model.py:200-210, written as free functions over the reference's
state_dict key names instead of as nn.Modules:

  stem ............ model.py:164-187   (conv3x3 + norm + ReLU chain, last one stride 2)
  stage input ..... model.py:137,145   (1x1 conv)
  hourglass ....... model.py:25-47     (pre-activation bottleneck ResBlocks :6-23,
                                        MaxPool2d(2,2), nearest up-sample, skip add)
  heads ........... model.py:54-65, 103-114
  decoder ......... model.py:79-97, 123-132  (delegated to torch ops here; numpy twin in
                                              decoder_ref.py)
  stage coupling .. model.py:204-208   (next input = cat[heatmaps, depthmaps, label_img])

It is differentiable (torch autograd on CPU), so it also provides reference gradients.
Used as: parity checker in tests/, smoke() checker, and bench.py's ``cpu_baseline`` ("port").
Never imported by the product package.

``forward(..., storage="bf16")`` (round 5) is the same evaluation ROUNDED WHERE THE bf16 ENGINE ROUNDS: every tensor
the engine stores between kernels is bfloat16 -- each conv output (after bias and the residual / skip add of its epilogue), each
normalised + ReLU'd conv operand, the up-sample + skip sums, the stage-input concat -- and so is the GRADIENT the engine stores for each of
them (the data-gradient convs and the norm / pool / up-sample backward kernels write bf16); conv weights enter the MFMA as bf16 while their
gradients stay fp32; sums, norm statistics and the decoder are fp32.  Implemented as one autograd function applied at those tensors
(round the value forward, round the accumulated gradient backward; autograd sums the gradients of a tensor's consumers BEFORE the
producer's backward, which is where the engine's norm / pool backward kernels add their `addend` and round once).  What it is for: a bf16
implementation is ~100 % away from the float64 gradient below the last heads on the trained fixture (bf16 storage of the gradient maps
meets the mean subtraction of every InstanceNorm backward), so float64 cannot tell a wrong bf16 hourglass / stem gradient kernel from a
right one; against THIS oracle the engine's per-group gradients have to agree to a few per cent (tests/test_trained_fixture_gpu.py).
"""
import torch
import torch.nn.functional as F

EPS = 1e-14


class _RoundBf16(torch.autograd.Function):
    """value -> bf16 -> back (forward), gradient -> bf16 -> back (backward); mode 1: forward only, 2: backward only"""

    @staticmethod
    def forward(ctx, x, mode):
        ctx.mode = mode
        return x if mode == 2 else x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return (g if ctx.mode == 1 else g.to(torch.bfloat16).to(g.dtype)), None


class _Storage:
    """where the bf16 engine rounds: q = a stored activation (value and gradient), qw = a conv weight as the MFMA sees it (value only),
    qg = an fp32 tensor whose gradient is stored in bf16 (the heads' output maps: the decoder's gradient goes to NHWC bf16)"""

    def __init__(self, storage):
        self.on = storage == "bf16"
        if storage not in ("fp32", "bf16"):
            raise ValueError("storage: 'fp32' or 'bf16'")

    def q(self, x):
        return _RoundBf16.apply(x, 0) if self.on else x

    def qw(self, w):
        return _RoundBf16.apply(w, 1) if self.on else w

    def qg(self, x):
        return _RoundBf16.apply(x, 2) if self.on else x


_FP32 = _Storage("fp32")


class RefConfig:
    def __init__(self, joints, stage=2, label_size=64, features=256, level=4, kernel_size=3,
                 norm_method="batch", heatmap_method="softmax"):
        self.joints, self.stage, self.label_size = joints, stage, label_size
        self.features, self.level, self.kernel_size = features, level, kernel_size
        self.norm_method, self.heatmap_method = norm_method, heatmap_method


def _norm(x, sd, prefix, cfg, training, bn_updates):
    wt, bs = sd[prefix + ".weight"], sd[prefix + ".bias"]
    if cfg.norm_method == "instance":
        # InstanceNorm2d(affine=True, track_running_stats=False): per-(b,c) statistics always
        return F.instance_norm(x, None, None, wt, bs, True, 0.1, 1e-5)
    rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    if training:
        rm2, rv2 = rm.clone(), rv.clone()
        y = F.batch_norm(x, rm2, rv2, wt, bs, True, 0.1, 1e-5)
        if bn_updates is not None:
            bn_updates[prefix] = (rm2, rv2)
        return y
    return F.batch_norm(x, rm, rv, wt, bs, False, 0.1, 1e-5)


# (ReLU in place on the norm's fresh output, like the reference's nn.ReLU(inplace=True), model.py:12,15,18,57-63,167-185: same values,
# one tensor less per layer -- with an out-of-place ReLU the oracle's inference ran at 0.78x the reference's speed, round-2 review)
def _conv(x, sd, prefix, stride=1, padding=0, st=_FP32, mfma=True):
    # (mfma=False: the stem's first conv, Cin = 1, runs on the vector ALU with fp32 weights)
    return F.conv2d(x, st.qw(sd[prefix + ".weight"]) if mfma else sd[prefix + ".weight"], sd[prefix + ".bias"], stride=stride, padding=padding)


def _resblock(x, sd, prefix, cfg, tr, bu, st=_FP32):
    # model.py:10-23 -- norm,ReLU,1x1 (F->F/2), norm,ReLU,kxk, norm,ReLU,1x1 (F/2->F); + x.
    # The hourglass ResBlocks always use kernel_size 3 (model.py:139 does not forward it).
    # (bf16 storage: every conv operand and every conv output is a stored tensor; the last conv's epilogue adds x before its one rounding)
    h = st.q(F.relu_(_norm(x, sd, prefix + ".conv.0", cfg, tr, bu)))
    h = st.q(_conv(h, sd, prefix + ".conv.2", st=st))
    h = st.q(F.relu_(_norm(h, sd, prefix + ".conv.3", cfg, tr, bu)))
    k = sd[prefix + ".conv.5.weight"].shape[-1]
    h = st.q(_conv(h, sd, prefix + ".conv.5", padding=k // 2, st=st))
    h = st.q(F.relu_(_norm(h, sd, prefix + ".conv.6", cfg, tr, bu)))
    h = _conv(h, sd, prefix + ".conv.8", st=st)
    return st.q(x + h)


def _hourglass(x, sd, prefix, level, cfg, tr, bu, st=_FP32):
    x = _resblock(x, sd, prefix + ".input_conv", cfg, tr, bu, st)       # model.py:39
    h = F.max_pool2d(x, 2, stride=2)                                    # model.py:40 (copies stored values: nothing to round)
    if level > 0:
        h = _hourglass(h, sd, prefix + ".inner", level - 1, cfg, tr, bu, st)
    else:
        h = _resblock(h, sd, prefix + ".inner", cfg, tr, bu, st)        # model.py:34
    h = _resblock(h, sd, prefix + ".output_conv", cfg, tr, bu, st)      # model.py:44
    h = F.interpolate(h, size=x.shape[2:])                              # model.py:45 (nearest)
    return st.q(h + x)


def _head(f, sd, prefix, cfg, tr, bu, st=_FP32):
    pad = cfg.kernel_size // 2
    h = f
    for i in (0, 3, 6):
        h = st.q(_conv(h, sd, "%s.conv.%d" % (prefix, i), padding=pad, st=st))
        h = st.q(F.relu_(_norm(h, sd, "%s.conv.%d" % (prefix, i + 1), cfg, tr, bu)))
    # the last conv writes the fp32 NCHW map the decoder reads; the gradient coming back from the decoder is stored in bf16
    return st.qg(_conv(h, sd, prefix + ".conv.9", padding=pad, st=st))


def decode_torch(z, D, L, m, w, grid, method):
    """Decoder in torch ops (differentiable); same arithmetic as decoder_ref.decode_forward."""
    B, J, H, W = z.shape
    if method == "softmax":
        p = F.softmax(w * z.reshape(B, J, -1), dim=2).reshape(B, J, H, W)
    else:
        r = F.relu(z) + EPS
        p = r / r.sum(dim=(2, 3), keepdim=True)
    u = (grid[0].view(1, 1, H, W) * p).sum(dim=(2, 3))
    v = (grid[1].view(1, 1, H, W) * p).sum(dim=(2, 3))
    mp = p * m
    d = (mp * (m * (D + L))).sum(dim=(2, 3)) / (mp.sum(dim=(2, 3)) + EPS)
    return p, torch.stack([u, v, d], dim=2)


def forward(sd, cfg, img, label_img, mask, training=True, bn_updates=None, storage="fp32"):
    """Returns list (len = cfg.stage) of (heatmaps, depthmaps, uvd) like model.py:200-210.  storage="bf16": rounded where the bf16 engine
    rounds (module docstring)."""
    st = _Storage(storage)
    pad = cfg.kernel_size // 2
    # ---- stem (model.py:164-187): indices 0,3,6,.. are convs, +1 norms
    n_stem = sum(1 for k in sd if k.startswith("conv.") and k.endswith(".weight") and sd[k].dim() == 4)
    f = img
    for i in range(n_stem):
        stride = 2 if i == n_stem - 1 else 1
        f = st.q(_conv(f, sd, "conv.%d" % (3 * i), stride=stride, padding=pad, st=st, mfma=i > 0))
        f = st.q(F.relu_(_norm(f, sd, "conv.%d" % (3 * i + 1), cfg, training, bn_updates)))
    results = []
    for s in range(cfg.stage):
        pre = "stages.%d" % s
        x = st.q(_conv(f, sd, pre + ".conv", st=st))
        feat = _hourglass(x, sd, pre + ".hourglass", cfg.level, cfg, training, bn_updates, st)
        z = _head(feat, sd, pre + ".plane_regression", cfg, training, bn_updates, st)
        D = _head(feat, sd, pre + ".depth_regression", cfg, training, bn_updates, st)
        w = sd.get(pre + ".plane_regression.w")
        p, uvd = decode_torch(z, D, label_img, mask, w, sd[pre + ".plane_regression.filter"],
                              cfg.heatmap_method)
        results.append((p, D, uvd))
        f = st.q(torch.cat([p, D, label_img], dim=1))                    # model.py:208 (the engine writes the concat once, as NHWC in the storage type)
    return results


def train_loss(results, uvd_t, heat_t=None, depth_t=None, alpha=1.0, lambda_h=1.0, lambda_d=0.01):
    """The 3-term multi-stage loss of /root/reference/train.py:195-205."""
    loss = 0
    for (p, D, uvd) in results:
        uvd_loss = torch.mean(torch.sum((uvd - uvd_t) ** 2, dim=2))
        if heat_t is not None:
            hl = lambda_h * torch.mean(torch.sum((p - heat_t) ** 2, dim=(2, 3)))
            dl = lambda_d * torch.mean(torch.sum((D - depth_t) ** 2, dim=(2, 3)))
        else:
            hl = dl = 0.0
        loss = loss + alpha * uvd_loss + (1 - alpha) * (hl + dl)
    return loss
