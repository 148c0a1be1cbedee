"""Thin tensor-level wrappers over the C ABI (include/pwr.h), one per kernel entry point.

Used by the op-level parity tests and for debugging; the network engine (csrc/engine.cpp) calls the
same entry points from C++ without going through Python.  Activations are NHWC torch tensors
([B,H,W,C], float32 or bfloat16); everything must live on the GPU.
"""
import struct

import ctypes

import torch

from . import _lib

F32, BF16 = 0, 1


def _dt(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError("activations must be float32 or bfloat16, got %s" % t.dtype)


def _s(t):
    return _lib.stream_ptr(t.device)


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.PwrError("pwr kernels need GPU tensors")
    assert t.is_contiguous()
    return t.data_ptr()


def pack_descs(entries, device):
    """entries: list of (src_off_floats, dst_off_bytes, Cout, Cin, ksize, kind, dtype[, order]) -> device tensor."""
    l = _lib.lib()
    buf = bytearray()
    for ent in entries:
        (src, dst, cout, cin, k, kind, dtype), order = ent[:7], (ent[7] if len(ent) > 7 else 0)
        rows, kdim = (cout, cin) if kind == 0 else (cin, cout)
        ke = 32 if dtype == BF16 else 16
        buf += struct.pack("<qqiiiiiiii", src, dst, cout, cin, k, kind, l.pwr_conv_out_pad(rows), (kdim + ke - 1) // ke,
                           dtype, order)
    return torch.frombuffer(buf, dtype=torch.uint8).clone().to(device)


class FragPack:
    """A weight pack in the fragment order of the weight-stationary 128 -> 128 3x3 conv (csrc/conv_wstat.hip): the C ABI takes its address with
    bit 0 set, and only for the shapes that kernel takes."""

    def __init__(self, t):
        self.t = t

    def data_ptr(self):
        return self.t.data_ptr() | 1

    is_cuda = True

    def is_contiguous(self):
        return True


def pack_conv(weight, kind, dtype, frag=False):
    """weight: OIHW fp32 cuda tensor -> packed uint8 buffer for pwr_conv_fwd.  frag: the fragment-order pack (bf16 128 x 128 3x3 only)."""
    l = _lib.lib()
    cout, cin, k, _ = weight.shape
    if frag and not (dtype == BF16 and cout == 128 and cin == 128 and k == 3 and kind == 0):
        raise ValueError("fragment-order packs exist for the forward (kind 0) bf16 128 x 128 3x3 stride-1 weights only (include/pwr.h)")
    nbytes = l.pwr_conv_pack_bytes(cout, cin, k, kind, dtype)
    packs = torch.empty(nbytes, dtype=torch.uint8, device=weight.device)
    flat = weight.contiguous().float().view(-1)
    descs = pack_descs([(0, 0, cout, cin, k, kind, dtype, 1 if frag else 0)], weight.device)
    _lib.check(l.pwr_pack_weights(_p(flat), _p(packs), _p(descs), 1, _s(weight)), "pwr_pack_weights")
    return FragPack(packs) if frag else packs


def conv_fwd(x, wpack, cout, ksize, stride=1, bias=None, norm=None, relu_in=True, residual=None, mode=0,
             nhwc_out=True, nchw_out=False):
    """norm: the [4,B,Cin] state tensor of norm_stats (mean, rstd, scale, beta) or None."""
    l = _lib.lib()
    B, H, W, Cin = x.shape
    pad = ksize // 2
    if mode == 0:
        Ho, Wo = (H + 2 * pad - ksize) // stride + 1, (W + 2 * pad - ksize) // stride + 1
    else:
        Ho, Wo = 2 * H, 2 * W
    y = torch.empty(B, Ho, Wo, cout, dtype=x.dtype, device=x.device) if nhwc_out else None
    yn = torch.empty(B, cout, Ho, Wo, dtype=torch.float32, device=x.device) if nchw_out else None
    _lib.check(l.pwr_conv_fwd(_p(x), _p(wpack), _p(bias), _p(norm), int(relu_in), _p(residual), _p(y), _p(yn),
                              B, H, W, Cin, cout, ksize, stride, mode, _dt(x), _s(x)), "pwr_conv_fwd")
    return y, yn


def conv_fwd_nchw_pair(xa, wa, xb, wb, cout, ksize, bias_a=None, bias_b=None, norm_a=None, norm_b=None, relu_in=True):
    """two conv_fwd(..., nhwc_out=False, nchw_out=True) of one shape as ONE launch (the heads' last convs); None where the shape has no such
    launch (PWR_EUNSUPPORTED)"""
    l = _lib.lib()
    B, H, W, Cin = xa.shape
    ya = torch.empty(B, cout, H, W, dtype=torch.float32, device=xa.device)
    yb = torch.empty(B, cout, H, W, dtype=torch.float32, device=xa.device)
    rc = l.pwr_conv_fwd_nchw_pair(_p(xa), _p(wa), _p(bias_a), _p(norm_a), _p(ya), _p(xb), _p(wb), _p(bias_b), _p(norm_b), _p(yb),
                                  int(relu_in), B, H, W, Cin, cout, ksize, _dt(xa), _s(xa))
    if rc == -2:          # PWR_EUNSUPPORTED
        return None
    _lib.check(rc, "pwr_conv_fwd_nchw_pair")
    return ya, yb


def conv_wgrad(x, dy, cout_real, ksize, stride=1, norm=None, relu_in=True, splits=8, dw=None, cin_real=None):
    l = _lib.lib()
    B, H, W, Cin = x.shape
    Cout = dy.shape[-1]
    slab = torch.empty(l.pwr_conv_wgrad_slab_bytes(Cout, Cin, ksize, splits) // 4, dtype=torch.float32, device=x.device)
    acc = dw is not None
    cin_real = cin_real or Cin
    if dw is None:
        dw = torch.empty(cout_real, cin_real, ksize, ksize, dtype=torch.float32, device=x.device)
    _lib.check(l.pwr_conv_wgrad(_p(x), _p(dy), _p(norm), int(relu_in), _p(slab), _p(dw), int(acc), B, H, W, Cin, cin_real,
                                Cout, cout_real, ksize, stride, splits, _dt(x), _s(x)), "pwr_conv_wgrad")
    return dw


class _WgradJob(ctypes.Structure):
    _fields_ = [("x", ctypes.c_void_p), ("dy", ctypes.c_void_p), ("in_norm", ctypes.c_void_p), ("dw", ctypes.c_void_p), ("H", ctypes.c_int),
                ("W", ctypes.c_int), ("Cin", ctypes.c_int), ("cin_real", ctypes.c_int), ("Cout", ctypes.c_int), ("cout_real", ctypes.c_int),
                ("ksize", ctypes.c_int), ("relu_in", ctypes.c_int)]


def conv_wgrad_group(jobs):
    """jobs: list of (x [B,H,W,Cin], dy [B,H,W,Cout], ksize, norm state or None): the weight gradients of all of them in one grouped
    launch (pwr_conv_wgrad_group).  Returns the list of dw [Cout,Cin,k,k] fp32."""
    l = _lib.lib()
    B = jobs[0][0].shape[0]
    arr = (_WgradJob * len(jobs))()
    dws = []
    for i, (x, dy, k, norm) in enumerate(jobs):
        _, H, W, Cin = x.shape
        Cout = dy.shape[-1]
        dw = torch.empty(Cout, Cin, k, k, dtype=torch.float32, device=x.device)
        dws.append(dw)
        arr[i] = _WgradJob(_p(x), _p(dy), _p(norm), _p(dw), H, W, Cin, Cin, Cout, Cout, k, 1 if norm is not None else 0)
    nbytes = l.pwr_conv_wgrad_group_slab_bytes(arr, len(jobs), B)
    if not nbytes:
        raise _lib.PwrError("pwr_conv_wgrad_group: unsupported job list")
    slab = torch.empty(nbytes // 4, dtype=torch.float32, device=jobs[0][0].device)
    _lib.check(l.pwr_conv_wgrad_group(arr, len(jobs), _p(slab), B, _dt(jobs[0][0]), _s(jobs[0][0])), "pwr_conv_wgrad_group")
    return dws


def stem_conv_fwd(img, w, bias, dtype):
    l = _lib.lib()
    B, S = img.shape[0], img.shape[-1]
    C0, k = w.shape[0], w.shape[-1]
    y = torch.empty(B, S, S, C0, dtype=dtype, device=img.device)
    _lib.check(l.pwr_stem_conv_fwd(_p(img), _p(w), _p(bias), _p(y), B, S, C0, k, _dt(y), _s(img)), "pwr_stem_conv_fwd")
    return y


def stem_conv_wgrad(img, dy, ksize):
    l = _lib.lib()
    B, S, _, C0 = dy.shape
    nb = l.pwr_stem_conv_wgrad_blocks(B, S)
    slab = torch.empty(nb * C0 * ksize * ksize, dtype=torch.float32, device=dy.device)
    dw = torch.empty(C0, 1, ksize, ksize, dtype=torch.float32, device=dy.device)
    _lib.check(l.pwr_stem_conv_wgrad(_p(img), _p(dy), _p(slab), _p(dw), 0, B, S, C0, ksize, _dt(dy), _s(dy)),
               "pwr_stem_conv_wgrad")
    return dw


def catconv_fwd(pmap, dmap, label, w, bias, dtype):
    l = _lib.lib()
    B, J, P, _ = pmap.shape
    Fo = w.shape[0]
    y = torch.empty(B, P, P, Fo, dtype=dtype, device=pmap.device)
    _lib.check(l.pwr_catconv_fwd(_p(pmap), _p(dmap), _p(label), _p(w), _p(bias), _p(y), B, P * P, J, Fo, _dt(y), _s(y)),
               "pwr_catconv_fwd")
    return y


def catconv_dgrad(dy, w, J):
    l = _lib.lib()
    B, P, _, Fo = dy.shape
    gp = torch.empty(B, J, P, P, dtype=torch.float32, device=dy.device)
    gd = torch.empty_like(gp)
    _lib.check(l.pwr_catconv_dgrad(_p(dy), _p(w), _p(gp), _p(gd), B, P * P, J, Fo, _dt(dy), _s(dy)), "pwr_catconv_dgrad")
    return gp, gd


def catconv_wgrad(pmap, dmap, label, dy):
    l = _lib.lib()
    B, J, P, _ = pmap.shape
    Fo = dy.shape[-1]
    nb = l.pwr_catconv_wgrad_blocks(B, P * P)
    slab = torch.empty(nb * (2 * J + 2) * Fo, dtype=torch.float32, device=dy.device)
    dw = torch.empty(Fo, 2 * J + 1, 1, 1, dtype=torch.float32, device=dy.device)
    db = torch.empty(Fo, dtype=torch.float32, device=dy.device)
    _lib.check(l.pwr_catconv_wgrad(_p(pmap), _p(dmap), _p(label), _p(dy), _p(slab), _p(dw), _p(db), 0, B, P * P, J, Fo,
                                   _dt(dy), _s(dy)), "pwr_catconv_wgrad")
    return dw, db


def norm_stats(y, gamma, beta, mode=0, running_mean=None, running_var=None, eps=1e-5, momentum=0.1):
    l = _lib.lib()
    B, H, W, C = y.shape
    dev = y.device
    partial = torch.zeros(l.pwr_norm_partial_bytes(B, H * W, C) // 4, dtype=torch.float32, device=dev)   # hand-off counters must start at 0
    state = torch.empty(4, B, C, dtype=torch.float32, device=dev)
    _lib.check(l.pwr_norm_stats(_p(y), _p(gamma), _p(beta), _p(running_mean), _p(running_var), _p(partial), _p(state),
                                B, H * W, C, mode, eps, momentum, _dt(y), _s(y)), "pwr_norm_stats")
    return state   # [4,B,C]: mean, rstd, scale = gamma*rstd, beta


def norm_bwd(g, y, state, relu=True, addend=None, mode=0):
    l = _lib.lib()
    B, H, W, C = y.shape
    dev = y.device
    partial = torch.empty(l.pwr_norm_partial_bytes(B, H * W, C) // 4, dtype=torch.float32, device=dev)
    S1 = torch.empty(B, C, dtype=torch.float32, device=dev)
    S2 = torch.empty_like(S1)
    dy = torch.empty_like(y)
    dgamma = torch.empty(C, dtype=torch.float32, device=dev)
    dbeta = torch.empty_like(dgamma)
    _lib.check(l.pwr_norm_bwd(_p(g), _p(y), _p(state), _p(partial), _p(S1), _p(S2),
                              _p(addend), _p(dy), _p(dgamma), _p(dbeta), 0, int(relu), B, H * W, C, mode, _dt(y), _s(y)),
               "pwr_norm_bwd")
    return dy, dgamma, dbeta


def maxpool_fwd(x):
    l = _lib.lib()
    B, H, W, C = x.shape
    y = torch.empty(B, H // 2, W // 2, C, dtype=x.dtype, device=x.device)
    _lib.check(l.pwr_maxpool_fwd(_p(x), _p(y), B, H, W, C, _dt(x), _s(x)), "pwr_maxpool_fwd")
    return y


def maxpool_bwd(x, dh, addend=None):
    l = _lib.lib()
    B, H, W, C = x.shape
    dx = torch.empty_like(x)
    _lib.check(l.pwr_maxpool_bwd(_p(x), _p(dh), _p(addend), _p(dx), B, H, W, C, _dt(x), _s(x)), "pwr_maxpool_bwd")
    return dx


def upsample_add(h, skip):
    l = _lib.lib()
    B, Hi, Wi, C = h.shape
    _, Ho, Wo, _ = skip.shape
    out = torch.empty_like(skip)
    _lib.check(l.pwr_upsample_add_fwd(_p(h), _p(skip), _p(out), B, Hi, Wi, Ho, Wo, C, _dt(h), _s(h)), "pwr_upsample_add_fwd")
    return out


def upsample_bwd(dout, Hi, Wi):
    l = _lib.lib()
    B, Ho, Wo, C = dout.shape
    dh = torch.empty(B, Hi, Wi, C, dtype=dout.dtype, device=dout.device)
    _lib.check(l.pwr_upsample_bwd(_p(dout), _p(dh), B, Hi, Wi, Ho, Wo, C, _dt(dout), _s(dout)), "pwr_upsample_bwd")
    return dh


def nchw_to_nhwc_pad(src, Jp, dtype):
    l = _lib.lib()
    B, J, P, _ = src.shape
    dst = torch.empty(B, P, P, Jp, dtype=dtype, device=src.device)
    _lib.check(l.pwr_nchw_to_nhwc_pad(_p(src), _p(dst), B, J, P * P, Jp, _dt(dst), _s(src)), "pwr_nchw_to_nhwc_pad")
    return dst


def cat_to_nhwc(pmap, dmap, label, dtype):
    l = _lib.lib()
    B, J, P, _ = pmap.shape
    Cp = (2 * J + 1 + 7) // 8 * 8
    dst = torch.empty(B, P, P, Cp, dtype=dtype, device=pmap.device)
    _lib.check(l.pwr_cat_to_nhwc(_p(pmap), _p(dmap), _p(label), _p(dst), B, J, P * P, Cp, _dt(dst), _s(dst)), "pwr_cat_to_nhwc")
    return dst


def nhwc_to_cat_grad(src, J):
    l = _lib.lib()
    B, P, _, Cp = src.shape
    gp = torch.empty(B, J, P, P, dtype=torch.float32, device=src.device)
    gd = torch.empty_like(gp)
    _lib.check(l.pwr_nhwc_to_cat_grad(_p(src), _p(gp), _p(gd), B, J, P * P, Cp, _dt(src), _s(src)), "pwr_nhwc_to_cat_grad")
    return gp, gd


def colsum_nhwc(x):
    l = _lib.lib()
    C = x.shape[-1]
    M = x.numel() // C
    slab = torch.empty(l.pwr_colsum_blocks(M) * C, dtype=torch.float32, device=x.device)
    out = torch.empty(C, dtype=torch.float32, device=x.device)
    _lib.check(l.pwr_colsum_nhwc(_p(x), _p(slab), _p(out), M, C, 0, _dt(x), _s(x)), "pwr_colsum_nhwc")
    return out


def planesum_nchw(x):
    l = _lib.lib()
    B, J, P, _ = x.shape
    part = torch.empty(B * J, dtype=torch.float32, device=x.device)
    out = torch.empty(J, dtype=torch.float32, device=x.device)
    _lib.check(l.pwr_planesum_nchw(_p(x), _p(part), _p(out), B, J, P * P, 0, _s(x)), "pwr_planesum_nchw")
    return out


def resblock_small_supported(H, W, C, norm_mode, dtype):
    return bool(_lib.lib().pwr_resblock_small_supported(H, W, C, norm_mode, dtype))


def resblock_fwd_small(x, packs, biases, gammas, betas, eps=1e-5):
    """One-launch ResBlock forward on a small map.  packs/biases/gammas/betas: 3-tuples for (a, b, c) = (1x1 C->C/2, 3x3, 1x1
    C/2->C); packs of kind 0.  Returns out, t1, t2, (state_a, state_b, state_c)."""
    l = _lib.lib()
    B, H, W, C = x.shape
    dev = x.device
    t1 = torch.empty(B, H, W, C // 2, dtype=x.dtype, device=dev)
    t2 = torch.empty_like(t1)
    out = torch.empty_like(x)
    st = [torch.empty(4, B, c, dtype=torch.float32, device=dev) for c in (C, C // 2, C // 2)]
    _lib.check(l.pwr_resblock_fwd_small(_p(x), _p(t1), _p(t2), _p(out), _p(packs[0]), _p(packs[1]), _p(packs[2]),
                                        _p(biases[0]), _p(biases[1]), _p(biases[2]), _p(gammas[0]), _p(betas[0]), _p(gammas[1]),
                                        _p(betas[1]), _p(gammas[2]), _p(betas[2]), _p(st[0]), _p(st[1]), _p(st[2]), B, H, W, C, eps,
                                        _dt(x), _s(x)), "pwr_resblock_fwd_small")
    return out, t1, t2, st


def resblock_fwd_small_x(xmode, xa, xh, packs, biases, gammas, betas, eps=1e-5):
    """resblock_fwd_small with its input's producer fused into the load: xmode 1: x = maxpool2x2(xa); xmode 2: x = upsample(xh) + xa.
    Returns x (as the kernel wrote it), out, t1, t2, states."""
    l = _lib.lib()
    B, Ha, Wa, C = xa.shape
    H, W = (Ha // 2, Wa // 2) if xmode == 1 else (Ha, Wa)
    dev = xa.device
    x = torch.full((B, H, W, C), float("nan"), dtype=xa.dtype, device=dev)
    t1 = torch.empty(B, H, W, C // 2, dtype=xa.dtype, device=dev)
    t2 = torch.empty_like(t1)
    out = torch.empty_like(x)
    st = [torch.empty(4, B, c, dtype=torch.float32, device=dev) for c in (C, C // 2, C // 2)]
    _lib.check(l.pwr_resblock_fwd_small_x(int(xmode), _p(xa), _p(xh), _p(x), _p(t1), _p(t2), _p(out), _p(packs[0]), _p(packs[1]), _p(packs[2]),
                                          _p(biases[0]), _p(biases[1]), _p(biases[2]), _p(gammas[0]), _p(betas[0]), _p(gammas[1]),
                                          _p(betas[1]), _p(gammas[2]), _p(betas[2]), _p(st[0]), _p(st[1]), _p(st[2]), B, H, W, C, eps,
                                          _dt(xa), _s(xa)), "pwr_resblock_fwd_small_x")
    return x, out, t1, t2, st


def resblock_bwd_small(gout, x, t1, t2, packs_d, states):
    """One-launch ResBlock backward.  packs_d: kind-1 packs of (a, b, c).  Returns dx, dt1, dt2, (sums_a, sums_b, sums_c)."""
    l = _lib.lib()
    B, H, W, C = x.shape
    dev = x.device
    dx, dt1, dt2 = torch.empty_like(x), torch.empty_like(t1), torch.empty_like(t2)
    sums = [torch.empty(B, 2, c, dtype=torch.float32, device=dev) for c in (C, C // 2, C // 2)]
    bsum = torch.empty(B, C, dtype=torch.float32, device=dev)
    _lib.check(l.pwr_resblock_bwd_small(_p(gout), _p(x), _p(t1), _p(t2), _p(dx), _p(dt1), _p(dt2), _p(packs_d[2]), _p(packs_d[1]),
                                        _p(packs_d[0]), _p(states[0]), _p(states[1]), _p(states[2]), _p(sums[0]), _p(sums[1]),
                                        _p(sums[2]), _p(bsum), B, H, W, C, _dt(x), _s(x)), "pwr_resblock_bwd_small")
    pg = [torch.empty(c, dtype=torch.float32, device=dev) for c in (C, C, C // 2, C // 2, C // 2, C // 2, C)]
    _lib.check(l.pwr_resblock_param_grads(_p(sums[0]), _p(sums[1]), _p(sums[2]), _p(bsum), *[_p(t) for t in pg], B, C, _s(x)),
               "pwr_resblock_param_grads")
    return dx, dt1, dt2, sums, pg   # pg = dgamma_a, dbeta_a, dgamma_b, dbeta_b, dgamma_c, dbeta_c, dbias_c


def resblock_bwd_small_x(x, t1, t2, packs_d, states, gout=None, up_src=None, pool_a=None, pool_addend=None):
    """pwr_resblock_bwd_small_x: the one-launch ResBlock backward with the up-sample backward fused into its load (up_src [B,2H,2W,C]: gout is
    computed and returned) and / or the max-pool backward fused into its store (pool_a, pool_addend [B,2H,2W,C]: returns pool_dst).
    Returns dx, dt1, dt2, sums, gout, pool_dst, bias_sums."""
    l = _lib.lib()
    B, H, W, C = x.shape
    dev = x.device
    dx, dt1, dt2 = torch.empty_like(x), torch.empty_like(t1), torch.empty_like(t2)
    sums = [torch.empty(B, 2, c, dtype=torch.float32, device=dev) for c in (C, C // 2, C // 2)]
    bsum = torch.empty(B, C, dtype=torch.float32, device=dev)
    if up_src is not None:
        gout = torch.full_like(x, float("nan"))
    pool_dst = torch.full_like(pool_a, float("nan")) if pool_a is not None else None
    _lib.check(l.pwr_resblock_bwd_small_x(_p(up_src), _p(pool_a), _p(pool_addend), _p(pool_dst), _p(gout), _p(x), _p(t1), _p(t2), _p(dx), _p(dt1),
                                          _p(dt2), _p(packs_d[2]), _p(packs_d[1]), _p(packs_d[0]), _p(states[0]), _p(states[1]), _p(states[2]),
                                          _p(sums[0]), _p(sums[1]), _p(sums[2]), _p(bsum), B, H, W, C, _dt(x), _s(x)), "pwr_resblock_bwd_small_x")
    return dx, dt1, dt2, sums, gout, pool_dst, bsum


def conv_stats_chunks(H, W, Cin, Cout, ksize, stride=1, mode=0, dtype=BF16):
    return _lib.lib().pwr_conv_stats_chunks(H, W, Cin, Cout, ksize, stride, mode, dtype)


def conv_fwd_stats(x, wpack, cout, ksize, stride=1, bias=None, norm=None, relu_in=True, residual=None, nb_y=None, nb_state=None,
                   nb_relu=True, mode=0):
    """conv_fwd + column statistics from the epilogue.  nb_y is None: forward statistics of the output (for the norm that
    follows); else: the norm-backward sums of the produced gradient w.r.t. (nb_y, nb_state).  Returns y, partial, chunks."""
    l = _lib.lib()
    B, H, W, Cin = x.shape
    pad = ksize // 2
    Ho, Wo = (H + 2 * pad - ksize) // stride + 1, (W + 2 * pad - ksize) // stride + 1
    if mode == 1:       # the data gradient of a stride-2 conv: x is the gradient map, the output has twice its size
        Ho, Wo = 2 * H, 2 * W
    chunks = l.pwr_conv_stats_chunks(H, W, Cin, cout, ksize, stride, mode, _dt(x))
    if chunks <= 0:
        raise _lib.PwrError("conv shape does not support epilogue statistics")
    y = torch.empty(B, Ho, Wo, cout, dtype=x.dtype, device=x.device)
    partial = torch.full((B * chunks, 3 if nb_y is None else 2, cout), float("nan"), dtype=torch.float32, device=x.device)
    st, nbp = (partial, None) if nb_y is None else (None, partial)
    _lib.check(l.pwr_conv_fwd_stats(_p(x), _p(wpack), _p(bias), _p(norm), int(relu_in), _p(residual), _p(y), B, H, W, Cin, cout, ksize,
                                    stride, mode, _p(st), _p(nb_y), _p(nb_state), _p(nbp), int(nb_relu), _dt(x), _s(x)), "pwr_conv_fwd_stats")
    return y, partial, chunks


def conv_fwd_stats_pair(xa, wa, xb, wb, cout, ksize, bias_a=None, bias_b=None, norm_a=None, norm_b=None, relu_in=True):
    """Two conv_fwd_stats (forward statistics form) of one shape in one launch.  Returns (ya, partial_a), (yb, partial_b), chunks."""
    l = _lib.lib()
    B, H, W, Cin = xa.shape
    chunks = l.pwr_conv_stats_chunks(H, W, Cin, cout, ksize, 1, 0, _dt(xa))
    if chunks <= 0:
        raise _lib.PwrError("conv shape does not support epilogue statistics")
    ya, yb = (torch.empty(B, H, W, cout, dtype=xa.dtype, device=xa.device) for _ in range(2))
    pa, pb = (torch.full((B * chunks, 3, cout), float("nan"), dtype=torch.float32, device=xa.device) for _ in range(2))
    _lib.check(l.pwr_conv_fwd_stats_pair(_p(xa), _p(wa), _p(bias_a), _p(norm_a), _p(ya), _p(pa), _p(xb), _p(wb), _p(bias_b), _p(norm_b), _p(yb),
                                         _p(pb), int(relu_in), B, H, W, Cin, cout, ksize, _dt(xa), _s(xa)), "pwr_conv_fwd_stats_pair")
    return (ya, pa), (yb, pb), chunks


def conv_dgrad_stats_pair(dya, wa, nb_y_a, nb_state_a, dyb, wb, nb_y_b, nb_state_b, cout, ksize, nb_relu=True):
    """Two stride-1 data gradients with norm-backward sums (conv_fwd_stats with nb_y) of one shape in one launch.
    Returns (dxa, partial_a), (dxb, partial_b), chunks."""
    l = _lib.lib()
    B, H, W, Cin = dya.shape
    chunks = l.pwr_conv_stats_chunks(H, W, Cin, cout, ksize, 1, 0, _dt(dya))
    if chunks <= 0:
        raise _lib.PwrError("conv shape does not support epilogue statistics")
    xa, xb = (torch.empty(B, H, W, cout, dtype=dya.dtype, device=dya.device) for _ in range(2))
    pa, pb = (torch.full((B * chunks, 2, cout), float("nan"), dtype=torch.float32, device=dya.device) for _ in range(2))
    _lib.check(l.pwr_conv_dgrad_stats_pair(_p(dya), _p(wa), _p(xa), _p(nb_y_a), _p(nb_state_a), _p(pa), _p(dyb), _p(wb), _p(xb), _p(nb_y_b),
                                           _p(nb_state_b), _p(pb), int(nb_relu), B, H, W, Cin, cout, ksize, _dt(dya), _s(dya)), "pwr_conv_dgrad_stats_pair")
    return (xa, pa), (xb, pb), chunks


def conv_dgrad_fold_stats_pair(ga, wa, nb_y_a, nb_state_a, fb_y_a, fb_state_a, fb_partial_a, gb, wb, nb_y_b, nb_state_b, fb_y_b, fb_state_b,
                               fb_partial_b, fb_pchunks, cout, ksize, fb_relu=True, nb_relu=True):
    """conv_dgrad_stats_pair on RAW gradients: the norm backward of (fb_y, fb_state) with the sums of fb_partial runs in the staging.
    Returns (dxa, partial_a, dya), (dxb, partial_b, dyb), chunks -- dy* = what norm_bwd_fold would have written."""
    l = _lib.lib()
    B, H, W, Cin = ga.shape
    chunks = l.pwr_conv_stats_chunks(H, W, Cin, cout, ksize, 1, 0, _dt(ga))
    if chunks <= 0:
        raise _lib.PwrError("conv shape does not support epilogue statistics")
    xa, xb = (torch.empty(B, H, W, cout, dtype=ga.dtype, device=ga.device) for _ in range(2))
    da, db = (torch.full((B, H, W, Cin), float("nan"), dtype=ga.dtype, device=ga.device) for _ in range(2))
    pa, pb = (torch.full((B * chunks, 2, cout), float("nan"), dtype=torch.float32, device=ga.device) for _ in range(2))
    _lib.check(l.pwr_conv_dgrad_fold_stats_pair(_p(ga), _p(wa), _p(xa), _p(nb_y_a), _p(nb_state_a), _p(pa), _p(fb_y_a), _p(fb_state_a), _p(fb_partial_a), _p(da),
                                                _p(gb), _p(wb), _p(xb), _p(nb_y_b), _p(nb_state_b), _p(pb), _p(fb_y_b), _p(fb_state_b), _p(fb_partial_b), _p(db),
                                                int(fb_pchunks), int(fb_relu), int(nb_relu), B, H, W, Cin, cout, ksize, _dt(ga), _s(ga)),
               "pwr_conv_dgrad_fold_stats_pair")
    return (xa, pa, da), (xb, pb, db), chunks


def conv_wgrad_pair(xa, dya, xb, dyb, norm_a=None, norm_b=None, relu_in=True, splits=42):
    """Two 3x3 stride-1 weight gradients of one geometry in one launch + one reduce (pwr_conv_wgrad_pair).  Returns dwa, dwb."""
    l = _lib.lib()
    B, H, W, Cin = xa.shape
    Cout = dya.shape[-1]
    slab = torch.empty(2 * l.pwr_conv_wgrad_slab_bytes(Cout, Cin, 3, splits) // 4, dtype=torch.float32, device=xa.device)
    dwa, dwb = (torch.empty(Cout, Cin, 3, 3, dtype=torch.float32, device=xa.device) for _ in range(2))
    _lib.check(l.pwr_conv_wgrad_pair(_p(xa), _p(dya), _p(norm_a), _p(dwa), _p(xb), _p(dyb), _p(norm_b), _p(dwb), int(relu_in), _p(slab), B, H, W,
                                     Cin, Cout, splits, _dt(xa), _s(xa)), "pwr_conv_wgrad_pair")
    return dwa, dwb


def norm_finalize_partial(partial, chunks, gamma, beta, B, HW, mode=0, running_mean=None, running_var=None, eps=1e-5, momentum=0.1):
    l = _lib.lib()
    C = gamma.numel()
    state = torch.empty(4, B, C, dtype=torch.float32, device=partial.device)
    _lib.check(l.pwr_norm_finalize_partial(_p(partial), chunks, _p(gamma), _p(beta), _p(running_mean), _p(running_var),
                                           _p(state), B, HW, C, mode, eps, momentum, _s(partial)), "pwr_norm_finalize_partial")
    return state


def norm_bwd_from_partial_pair(ga, ya, state_a, partial_a, gb, yb, state_b, partial_b, chunks, relu=True):
    """Two norm_bwd_from_partial (instance norm, no addend) of one shape as two launches.  Returns (dya, dgamma_a, dbeta_a), (dyb, ...)."""
    l = _lib.lib()
    B, H, W, C = ya.shape
    dev = ya.device
    S1 = torch.empty(2, B, C, dtype=torch.float32, device=dev)
    S2 = torch.empty_like(S1)
    outs = []
    for y in (ya, yb):
        outs.append((torch.empty_like(y), torch.empty(C, dtype=torch.float32, device=dev), torch.empty(C, dtype=torch.float32, device=dev)))
    (dya, dga, dba), (dyb, dgb, dbb) = outs
    _lib.check(l.pwr_norm_bwd_from_partial_pair(_p(ga), _p(ya), _p(state_a), _p(partial_a), _p(dya), _p(dga), _p(dba), _p(gb), _p(yb), _p(state_b),
                                                _p(partial_b), _p(dyb), _p(dgb), _p(dbb), chunks, _p(S1), _p(S2), 0, int(relu), B, H * W, C, _dt(ya),
                                                _s(ya)), "pwr_norm_bwd_from_partial_pair")
    return outs


def norm_bwd_from_partial(g, y, state, partial, chunks, relu=True, addend=None, mode=0):
    l = _lib.lib()
    B, H, W, C = y.shape
    dev = y.device
    S1 = torch.empty(B, C, dtype=torch.float32, device=dev)
    S2 = torch.empty_like(S1)
    dy = torch.empty_like(y)
    dgamma = torch.empty(C, dtype=torch.float32, device=dev)
    dbeta = torch.empty_like(dgamma)
    _lib.check(l.pwr_norm_bwd_from_partial(_p(g), _p(y), _p(state), _p(partial), chunks, _p(S1), _p(S2), _p(addend), _p(dy), _p(dgamma),
                                           _p(dbeta), 0, int(relu), B, H * W, C, mode, _dt(y), _s(y)), "pwr_norm_bwd_from_partial")
    return dy, dgamma, dbeta


def norm_bwd_fold(g, y, state, partial, chunks, relu=True, addend=None, pair=None):
    """Round 6: pwr_norm_bwd_from_partial as ONE launch on the stream (the apply step sums the slab rows itself) + the parameter sums as a
    launch of their own (pwr_norm_bwd_apply_from_partial, pwr_norm_bwd_params_from_partial).  pair = (g_b, y_b, state_b, partial_b): a second
    tensor of the same shape in the same launches.  Returns (dy, dgamma, dbeta) or, with `pair`, two such tuples."""
    l = _lib.lib()
    B, H, W, C = y.shape
    dev = y.device
    outs = [(torch.empty_like(y), torch.empty(C, dtype=torch.float32, device=dev), torch.empty(C, dtype=torch.float32, device=dev))]
    gb = yb = sb = pb = None
    if pair is not None:
        gb, yb, sb, pb = pair
        outs.append((torch.empty_like(yb), torch.empty(C, dtype=torch.float32, device=dev), torch.empty(C, dtype=torch.float32, device=dev)))
    _lib.check(l.pwr_norm_bwd_apply_from_partial(_p(g), _p(y), _p(state), _p(partial), _p(addend), _p(outs[0][0]), _p(gb), _p(yb), _p(sb), _p(pb),
                                                 _p(outs[1][0]) if pair is not None else None, chunks, int(relu), B, H * W, C, _dt(y), _s(y)),
               "pwr_norm_bwd_apply_from_partial")
    _lib.check(l.pwr_norm_bwd_params_from_partial(_p(partial), _p(outs[0][1]), _p(outs[0][2]), _p(pb), _p(outs[1][1]) if pair is not None else None,
                                                  _p(outs[1][2]) if pair is not None else None, chunks, 0, B, H * W, C, _s(y)),
               "pwr_norm_bwd_params_from_partial")
    return outs if pair is not None else outs[0]


def norm_finalize_partial_pair(partial_a, gamma_a, beta_a, partial_b, gamma_b, beta_b, chunks, B, HW, eps=1e-5):
    """two norm_finalize_partial (instance norm) of one shape as one launch.  Returns state_a, state_b."""
    l = _lib.lib()
    C = gamma_a.numel()
    sa, sb = (torch.empty(4, B, C, dtype=torch.float32, device=partial_a.device) for _ in range(2))
    _lib.check(l.pwr_norm_finalize_partial_pair(_p(partial_a), _p(gamma_a), _p(beta_a), _p(sa), _p(partial_b), _p(gamma_b), _p(beta_b), _p(sb), chunks, B, HW,
                                                C, eps, _s(partial_a)), "pwr_norm_finalize_partial_pair")
    return sa, sb


def norm_bwd_params_group(jobs, B):
    """jobs: list of (partial, chunks, HW, C): the parameter gradients of several norm backwards in one launch (pwr_norm_bwd_params_group).
    Returns [(dgamma, dbeta), ...]."""
    import ctypes
    l = _lib.lib()

    class Job(ctypes.Structure):
        _fields_ = [("partial", ctypes.c_void_p), ("dgamma", ctypes.c_void_p), ("dbeta", ctypes.c_void_p), ("HW", ctypes.c_int), ("C", ctypes.c_int),
                    ("chunks", ctypes.c_int)]
    outs, arr = [], (Job * len(jobs))()
    for i, (partial, chunks, HW, C) in enumerate(jobs):
        dg, db = (torch.empty(C, dtype=torch.float32, device=partial.device) for _ in range(2))
        outs.append((dg, db))
        arr[i] = Job(partial.data_ptr(), dg.data_ptr(), db.data_ptr(), HW, C, chunks)
    _lib.check(l.pwr_norm_bwd_params_group(ctypes.cast(arr, ctypes.c_void_p), len(jobs), 0, B, _s(jobs[0][0])), "pwr_norm_bwd_params_group")
    return outs


def norm_apply(y, state, relu=True):
    """relu(norm(y)) as a tensor of y's dtype (pwr_norm_apply): the operand the convs / weight gradients build on load"""
    l = _lib.lib()
    B, H, W, C = y.shape
    out = torch.empty_like(y)
    _lib.check(l.pwr_norm_apply(_p(y), _p(state), _p(out), int(relu), B, H * W, C, _dt(y), _s(y)), "pwr_norm_apply")
    return out


def norm_stats_fused_src(src, xa, xh, gamma, beta, eps=1e-5):
    """src 1: y = maxpool2x2(xa); src 2: y = xa + nearest-upsample(xh); fused with the InstanceNorm statistics of y (pwr_norm_stats_fused_src).
    Returns (y, state) or None where the shape has no fused form."""
    l = _lib.lib()
    B, Ha, Wa, C = xa.shape
    H, W = (Ha // 2, Wa // 2) if src == 1 else (Ha, Wa)
    y = torch.empty(B, H, W, C, dtype=xa.dtype, device=xa.device)
    partial = torch.zeros(l.pwr_norm_partial_bytes(B, H * W, C) // 4, dtype=torch.float32, device=xa.device)
    state = torch.empty(4, B, C, dtype=torch.float32, device=xa.device)
    rc = l.pwr_norm_stats_fused_src(src, _p(xa), _p(xh), _p(y), _p(gamma), _p(beta), _p(partial), _p(state), B, H, W, C, eps, _dt(xa), _s(xa))
    if rc == -2:          # PWR_EUNSUPPORTED
        return None
    _lib.check(rc, "pwr_norm_stats_fused_src")
    return y, state
