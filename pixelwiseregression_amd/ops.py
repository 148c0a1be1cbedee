"""torch.autograd bindings of individual HIP kernels (through the C ABI of include/pwr.h).

Only device tensors are accepted: there is no CPU implementation behind these ops by design.
"""
import torch

from . import _lib

HEATMAP_METHODS = {"softmax": 0, "sum": 1}


def _require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.PwrError("pixelwiseregression_amd ops run on the GPU only (got a %s tensor); "
                                "there is no CPU fallback" % t.device)


def decode_forward(z, D, L, m, w, method):
    """Fused soft-argmax decoder forward (model.py:79-97 + 123-132). Returns (p, uvd)."""
    _require_cuda(z, D, L, m, w)
    B, J, P, P2 = z.shape
    assert P == P2 and D.shape == z.shape and L.shape == (B, 1, P, P) and m.shape == (B, 1, P, P)
    z, D, L, m = (t.contiguous().float() for t in (z, D, L, m))
    wv = w.contiguous().float().view(-1) if w is not None else None
    p = torch.empty_like(z)
    uvd = torch.empty(B, J, 3, device=z.device, dtype=torch.float32)
    l = _lib.lib()
    _lib.check(l.pwr_decode_fwd(_lib.ptr(z), _lib.ptr(D), _lib.ptr(L), _lib.ptr(m), _lib.ptr(wv), _lib.ptr(p),
                                _lib.ptr(uvd), B, J, P, HEATMAP_METHODS[method], _lib.stream_ptr(z.device)),
               "pwr_decode_fwd")
    return p, uvd


def decode_backward(p, z, D, L, m, w, uvd, gH, gD, gU, method):
    """Closed-form decoder backward. Returns (gz, gD_total, gw or None)."""
    _require_cuda(p, z, D, L, m, gU)
    B, J, P, _ = z.shape
    gz = torch.empty_like(z)
    gDo = torch.empty_like(z)
    wv = w.contiguous().float().view(-1) if w is not None else None
    gw_part = torch.empty(B * J, device=z.device, dtype=torch.float32) if w is not None else None
    gH = gH.contiguous().float() if gH is not None else None
    gD = gD.contiguous().float() if gD is not None else None
    gU = gU.contiguous().float()
    l = _lib.lib()
    s = _lib.stream_ptr(z.device)
    _lib.check(l.pwr_decode_bwd(_lib.ptr(p), _lib.ptr(z), _lib.ptr(D), _lib.ptr(L), _lib.ptr(m), _lib.ptr(wv),
                                _lib.ptr(uvd), _lib.ptr(gH), _lib.ptr(gD), _lib.ptr(gU), _lib.ptr(gz), _lib.ptr(gDo),
                                _lib.ptr(gw_part), B, J, P, HEATMAP_METHODS[method], s), "pwr_decode_bwd")
    gw = None
    if w is not None:
        gw = torch.empty(J, device=z.device, dtype=torch.float32)
        _lib.check(l.pwr_decode_gw_reduce(_lib.ptr(gw_part), _lib.ptr(gw), B, J, 0, s), "pwr_decode_gw_reduce")
        gw = gw.view_as(w)
    return gz, gDo, gw


class DecodeFn(torch.autograd.Function):
    """p, uvd = decode(z, D, L, m, w).  D's own gradient path (depth-map loss) stays in autograd;
    this node adds the decoder's contribution gd * p * m^2 / S."""

    @staticmethod
    def forward(ctx, z, D, L, m, w, method):
        z, D, L, m = (t.contiguous().float() for t in (z, D, L, m))
        p, uvd = decode_forward(z, D, L, m, w, method)
        ctx.method = method
        ctx.save_for_backward(p, z, D, L, m, w if w is not None else torch.empty(0, device=z.device), uvd)
        ctx.has_w = w is not None
        return p, uvd

    @staticmethod
    def backward(ctx, gp, guvd):
        p, z, D, L, m, w, uvd = ctx.saved_tensors
        w = w if ctx.has_w else None
        if guvd is None:
            guvd = torch.zeros_like(uvd)
        gz, gD, gw = decode_backward(p, z, D, L, m, w, uvd, gp, None, guvd, ctx.method)
        return gz, gD, None, None, gw, None


def decode(z, D, L, m, w, method="softmax"):
    return DecodeFn.apply(z, D, L, m, w, method)
