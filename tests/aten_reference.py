"""Test / A-B instrument, NOT part of the product package: the conv stack of a pixelwiseregression_amd.PixelwiseRegression
module evaluated through PyTorch-ROCm library ops (MIOpen / ATen) on the GPU, with the HIP decoder.  The tests and
tools/bench_aten.py use it to put the native engine next to "what stock PyTorch does on the same MI355X".
The product module has no backend switch: `model(...)` always runs the HIP engine.
"""
import torch
import torch.nn.functional as F

from pixelwiseregression_amd.ops import decode


def _norm(m, node, x):
    if m.norm_method == "instance":
        return F.instance_norm(x, None, None, node.weight, node.bias, True, 0.1, 1e-5)
    if m.training:
        node.num_batches_tracked += 1
    return F.batch_norm(x, node.running_mean, node.running_var, node.weight, node.bias, m.training, 0.1, 1e-5)


def _conv(node, x, stride=1):
    k = node.weight.shape[-1]
    return F.conv2d(x, node.weight, node.bias, stride=stride, padding=k // 2)


def _resblock(m, rb, x):
    c = rb.conv
    h = _conv(getattr(c, "2"), F.relu(_norm(m, getattr(c, "0"), x)))
    h = _conv(getattr(c, "5"), F.relu(_norm(m, getattr(c, "3"), h)))
    h = _conv(getattr(c, "8"), F.relu(_norm(m, getattr(c, "6"), h)))
    return x + h


def _hourglass(m, hg, x, level):
    x = _resblock(m, hg.input_conv, x)
    h = F.max_pool2d(x, 2, stride=2)
    h = _hourglass(m, hg.inner, h, level - 1) if level > 0 else _resblock(m, hg.inner, h)
    h = _resblock(m, hg.output_conv, h)
    return F.interpolate(h, size=x.shape[2:]) + x


def _head(m, seq, f):
    h = f
    for i in (0, 3, 6):
        h = F.relu(_norm(m, getattr(seq, str(i + 1)), _conv(getattr(seq, str(i)), h)))
    return _conv(getattr(seq, "9"), h)


def aten_forward(m, img, label_img, mask):
    if m._precision == "bf16" and not torch.is_autocast_enabled():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            return aten_forward(m, img, label_img, mask)
    f = img
    for i in range(m.n_stem):
        f = _conv(getattr(m.conv, str(3 * i)), f, stride=2 if i == m.n_stem - 1 else 1)
        f = F.relu(_norm(m, getattr(m.conv, str(3 * i + 1)), f))
    results = []
    for blk in m.stages:
        feat = _hourglass(m, blk.hourglass, _conv(blk.conv, f), m.level)
        z = _head(m, blk.plane_regression.conv, feat)
        D = _head(m, blk.depth_regression.conv, feat)
        w = blk.plane_regression.w if m.heatmap_method == "softmax" else None
        p, uvd = decode(z.float(), D.float(), label_img, mask, w, m.heatmap_method)
        results.append((p, D, uvd))
        f = torch.cat([p, D.float(), label_img], dim=1)
    return results
