"""GPU parity of every kernel entry point of include/pwr.h (through kernels.py -> ctypes -> C ABI) against the
same op evaluated with plain torch ops in float64 on the CPU -- the per-op restatement of what
/root/reference/model.py composes (Conv2d, InstanceNorm2d/BatchNorm2d+ReLU, MaxPool2d, nearest interpolate).

Tolerances.  fp32 path: 2e-5 of max|ref| (exact-fp32 MFMA, different summation order).  bf16 path, the reference fed the SAME
bf16-rounded operands (products of bf16 operands are exact in fp32; only the fp32 summation order and the output rounding differ):
  * kernels that write fp32 (weight gradients, the heads' NCHW maps, statistics): 1e-4 of max|ref|; 2e-3 when the kernel applies a norm + ReLU
    to its operand on the way (its fp32 fma result is rounded to bf16 where the float64 reference's is: an operand can land one bf16 ulp
    away, rarely) -- `fp32_out_tol`;
  * kernels that write bf16: per ELEMENT half a bf16 ulp of the reference value (2^-8 |ref|: the one rounding of the output) plus the same
    1e-4 / 2e-3 of max|ref| for the summation order / the operand roundings -- `assert_close_bf16_out`.
(Rounds 1 - 4 held every bf16 kernel to 1.5e-2 of max|ref|, which a dropped K step out of 80 or a missing halo column would have passed.)
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
DTYPES = [torch.float32, torch.bfloat16]


def fp32_out_tol(dtype, prologue=False):
    """bound (relative to max|ref|) for a kernel whose OUTPUT is fp32"""
    if dtype == torch.float32:
        return 2e-5
    return 2e-3 if prologue else 1e-4


def assert_close_out(got, ref, dtype, prologue=False, what=""):
    """a kernel whose output has the activation dtype: fp32 -> 2e-5 of max|ref|; bf16 -> per element 2^-8 |ref| + (1e-4 | 2e-3) max|ref|"""
    if dtype == torch.float32:
        return assert_close(got, ref, 2e-5, what)
    den = max(ref.abs().max().item(), 1e-6)
    bound = ref.abs() * 2.0 ** -8 + (2e-3 if prologue else 1e-4) * den
    err = (got - ref).abs()
    worst = (err - bound).max().item()
    assert worst <= 0, "%s: an element is %.3e over its bound (max err %.3e, max|ref| %.3e)" % (what, worst, err.max().item(), den)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float64) * scale


def q(x, dtype):
    """value actually seen by the kernel for a tensor stored in `dtype` (float64 carrier)."""
    return x.to(dtype).double()


def nhwc(x, dtype):   # NCHW float64 -> NHWC device tensor of dtype
    return x.permute(0, 2, 3, 1).contiguous().to(dtype).to(DEV)


def nchw(y):          # NHWC device -> NCHW float64 cpu
    return y.double().cpu().permute(0, 3, 1, 2).contiguous()


def apply_nr(xin, B, C, dtype):
    """random norm state [4,B,C] (mean, rstd, scale, beta) and the NR-transformed input the kernel should see"""
    mean, sc, beta = 0.5 * rnd(B, C, seed=5), 1 + 0.2 * rnd(B, C, seed=6), 0.3 * rnd(B, C, seed=7)
    st = torch.stack([mean, torch.ones(B, C, dtype=torch.float64), sc, beta]).float()
    m, s_, b_ = (t.float().double()[:, :, None, None] for t in (mean, sc, beta))
    return q(torch.relu((xin - m) * s_ + b_), dtype), st.contiguous().to(DEV)


def assert_close(got, ref, rel, what=""):
    err = (got - ref).abs().max().item()
    den = max(ref.abs().max().item(), 1e-6)
    assert err <= rel * den, "%s: max err %.3e vs max|ref| %.3e (rel %.2e > %.1e)" % (what, err, den, err / den, rel)


CONV_CASES = [
    # B, H, W, Cin, Cout, k, stride
    (2, 16, 16, 32, 32, 3, 1),
    (1, 8, 8, 16, 16, 3, 1),
    (2, 64, 64, 128, 128, 3, 1),
    (3, 5, 7, 64, 64, 3, 1),
    (2, 32, 32, 128, 128, 3, 2),
    (2, 16, 16, 128, 64, 1, 1),
    (2, 16, 16, 64, 128, 1, 1),
    (1, 2, 2, 64, 64, 3, 1),
    (1, 16, 16, 64, 256, 3, 1),
    (2, 128, 128, 32, 64, 3, 1),
    # LDS-patch kernel shapes (W % 32 == 0, H % 4 == 0, Cin in {32,64,128}), incl. several tiles per row / image
    (2, 32, 32, 64, 64, 3, 1),
    (3, 64, 64, 64, 32, 3, 1),
    (1, 128, 128, 64, 128, 3, 1),
    (2, 4, 32, 128, 128, 3, 1),
    (1, 8, 96, 32, 16, 3, 1),
    # 1x1 convs through the LDS-patch kernel (bf16, W % 32 == 0, H % 4 == 0): bottleneck shapes of the ResBlocks, the stage-input
    # conv with its zero-padded 32-channel concat, 32- / 64- / 128-wide output tiles
    (2, 64, 64, 128, 64, 1, 1),
    (2, 64, 64, 64, 128, 1, 1),
    (3, 32, 32, 128, 128, 1, 1),
    (2, 8, 64, 32, 128, 1, 1),
    (2, 32, 32, 128, 24, 1, 1),
    # small square maps (whole images per tile): ragged last tile, several tiles, 2 N tiles
    (3, 2, 2, 64, 64, 3, 1),
    (40, 2, 2, 64, 64, 3, 1),
    (3, 4, 4, 64, 64, 3, 1),
    (9, 4, 4, 64, 48, 3, 1),
    (1, 8, 8, 64, 64, 3, 1),
    (5, 8, 8, 64, 64, 3, 1),
    (1, 16, 16, 64, 64, 3, 1),
    (3, 16, 16, 64, 40, 3, 1),
    # --filter_size 5 / 7 of the reference's scripts (train.py:47 -> model.py:55-64, :165-182): 25 / 49 taps
    (2, 16, 16, 32, 32, 5, 1),
    (2, 32, 32, 128, 128, 5, 1),
    (2, 16, 16, 32, 64, 7, 1),
    (1, 64, 64, 128, 128, 7, 1),
    (3, 5, 9, 64, 16, 7, 1),
    (2, 32, 32, 128, 128, 5, 2),
    (2, 20, 20, 64, 128, 7, 2),
    # stride-2 3x3 on maps whose OUTPUT has W % 32 == 0, H % 4 == 0 (bf16: the LDS-patch kernel's parity-class form, the stem's last conv)
    (2, 64, 64, 128, 128, 3, 2),
    (1, 128, 128, 128, 128, 3, 2),
    (3, 64, 64, 64, 64, 3, 2),
    (2, 8, 64, 32, 16, 3, 2),
    (2, 16, 128, 128, 40, 3, 2),
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("prologue", [False, True])
def test_conv_forward(case, dtype, prologue):
    from pixelwiseregression_amd import kernels as K
    B, H, W, Cin, Cout, k, stride = case
    x = rnd(B, Cin, H, W, seed=1)
    w = rnd(Cout, Cin, k, k, seed=2, scale=(Cin * k * k) ** -0.5)
    bias = rnd(Cout, seed=3, scale=0.1)
    res = rnd(B, Cout, (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1, seed=4)
    xin = q(x, dtype)
    st = None
    if prologue:
        xin, st = apply_nr(xin, B, Cin, dtype)
    ref = F.conv2d(xin, q(w, dtype), bias.float().double(), stride=stride, padding=k // 2) + q(res, dtype)
    pack = K.pack_conv(w.float().to(DEV), 0, K.BF16 if dtype == torch.bfloat16 else K.F32)
    y, _ = K.conv_fwd(nhwc(x, dtype), pack, Cout, k, stride, bias=bias.float().to(DEV), norm=st, relu_in=True,
                      residual=nhwc(res, dtype))
    assert_close_out(nchw(y), ref, dtype, prologue, "conv fwd %s" % (case,))


@pytest.mark.parametrize("B,H,W", [(2, 64, 64), (1, 32, 64), (5, 8, 96), (3, 32, 32)])
@pytest.mark.parametrize("form", ["plain", "norm", "norm_stats", "dgrad_nbsums"])
def test_weight_stationary_conv_fragment_order_pack(B, H, W, form):
    """csrc/conv_wstat.hip (the 128 -> 128 3x3 stride-1 bf16 conv with the weights stationary in registers): the pack in the kernel's own
    fragment order (PackDesc::order 1, handed over with bit 0 of the address set -- what the engine does for the heads' layers) gives
    exactly the bytes of the standard pack, outputs and epilogue statistics, for every tile count from 4 up (1 .. several tiles per persistent
    workgroup, partial last ranges); and a fragment-order pack on a shape the kernel does not take is refused."""
    from pixelwiseregression_amd import kernels as K, _lib
    x = nhwc(rnd(B, 128, H, W, seed=41), torch.bfloat16)
    w = rnd(128, 128, 3, 3, seed=42, scale=(128 * 9) ** -0.5).float().to(DEV)
    bias = (rnd(128, seed=43) * 3).float().to(DEV)
    _, st = apply_nr(q(rnd(B, 128, H, W, seed=41), torch.bfloat16), B, 128, torch.bfloat16)
    outs = []
    if form == "dgrad_nbsums":
        # the data gradients with norm-backward sums stay on the patch kernel (their epilogue does not hide behind one wave per SIMD's
        # MFMAs: DESIGN.md section 4): a fragment-order pack is refused there too, loudly
        nby = nhwc(rnd(B, 128, H, W, seed=44), torch.bfloat16)
        with pytest.raises(_lib.PwrError):
            K.conv_fwd_stats(x, K.FragPack(K.pack_conv(w, 1, K.BF16)), 128, 3, 1, nb_y=nby, nb_state=st)     # (a TAGGED address is what is refused)
        with pytest.raises(ValueError):
            K.pack_conv(w, 1, K.BF16, frag=True)                     # (fragment order exists for forward packs only: include/pwr.h)
        y, part, _ = K.conv_fwd_stats(x, K.pack_conv(w, 1, K.BF16), 128, 3, 1, nb_y=nby, nb_state=st)
        assert float(y.float().abs().max()) > 0 and not torch.isnan(part).any()
        return
    for frag in (False, True):
        if False:
            pass
        elif form == "norm_stats":
            y, part, _ = K.conv_fwd_stats(x, K.pack_conv(w, 0, K.BF16, frag=frag), 128, 3, 1, bias=bias, norm=st)
            outs.append((y, part))
        else:
            y, _ = K.conv_fwd(x, K.pack_conv(w, 0, K.BF16, frag=frag), 128, 3, 1, bias=bias, norm=st if form == "norm" else None)
            outs.append((y,))
    for a, b_ in zip(*outs):
        assert float(a.float().abs().max()) > 0 and torch.equal(a, b_), float((a.float() - b_.float()).abs().max())
    with pytest.raises(_lib.PwrError):
        K.conv_fwd(x, K.pack_conv(w, 0, K.BF16, frag=True), 128, 3, 2, bias=bias)          # stride 2: not that kernel's shape
    with pytest.raises(_lib.PwrError):
        K.conv_fwd(x, K.pack_conv(w, 0, K.BF16, frag=True), 128, 3, 1, bias=bias, mode=1)  # transposed mode: never reads a fragment-order pack


@pytest.mark.parametrize("case", [(2, 64, 64, 128), (9, 64, 64, 128), (3, 36, 96, 128), (2, 64, 64, 64), (1, 128, 128, 64), (3, 36, 96, 64)])
@pytest.mark.parametrize("form", ["plain", "norm", "norm_stats"])
def test_weight_stationary_conv_against_float64(case, form):
    """csrc/conv_wstat.hip held to the float64 CPU conv DIRECTLY (round-5 review, weak 2: it reached the oracle only through bit-equality with
    the patch kernel): the 128-input-channel form with the pack in the kernel's own FRAGMENT order (what the engine hands over), the
    64-input-channel form (the stem's 64 -> 128 conv, standard pack), plain / with the norm + ReLU prologue / with the prologue and the
    forward-statistics epilogue; shapes with one tile per workgroup, with 288 tiles over the 256 persistent workgroups (ranges of one and two
    tiles: a partial last range) and a map whose tile rows do not divide the image evenly among samples.  Bounds: assert_close_bf16_out, like
    every other bf16 kernel; the statistics, finalised, against the float64 statistics of the float64 conv output."""
    from pixelwiseregression_amd import kernels as K
    B, H, W, Cin = case
    dtype = torch.bfloat16
    x, w, bias = rnd(B, Cin, H, W, seed=11), rnd(128, Cin, 3, 3, seed=12, scale=(Cin * 9) ** -0.5), rnd(128, seed=13) * 2
    xin, st = q(x, dtype), None
    if form != "plain":
        xin, st = apply_nr(xin, B, Cin, dtype)
    ref = F.conv2d(xin, q(w, dtype), bias.float().double(), padding=1)
    pack = K.pack_conv(w.float().to(DEV), 0, K.BF16, frag=(Cin == 128))
    if form == "norm_stats":
        y, partial, chunks = K.conv_fwd_stats(nhwc(x, dtype), pack, 128, 3, 1, bias=bias.float().to(DEV), norm=st)
        assert not torch.isnan(partial).any()
        gamma, beta = (1 + 0.2 * rnd(128, seed=14)).float(), (0.2 * rnd(128, seed=15)).float()
        state = K.norm_finalize_partial(partial, chunks, gamma.to(DEV), beta.to(DEV), B, H * W, mode=0).double().cpu()
        rq = q(ref, dtype)                                          # the stored tensor the statistics are of
        mean, var = rq.mean(dim=(2, 3)), rq.var(dim=(2, 3), unbiased=False)
        rstd = 1.0 / torch.sqrt(var + 1e-5)
        sd_ = torch.sqrt(var)
        assert float(((state[0] - mean).abs() / sd_).max()) <= 2e-3, "statistics: mean off by %.2e standard deviations" % float(((state[0] - mean).abs() / sd_).max())
        assert_close(state[1], rstd, 2e-3, "statistics: rstd")
        assert_close(state[2], rstd * gamma.double()[None], 2e-3, "statistics: scale")
        assert torch.equal(state[3], beta.double()[None].expand(B, 128))
    else:
        y, _ = K.conv_fwd(nhwc(x, dtype), pack, 128, 3, 1, bias=bias.float().to(DEV), norm=st)
    assert_close_out(nchw(y), ref, dtype, form != "plain", "weight-stationary conv %s %s" % (case, form))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("J", [14, 4, 21])
@pytest.mark.parametrize("H,Cin", [(32, 64), (64, 128), (16, 64)])
def test_conv_forward_nchw_out(dtype, J, H, Cin):
    from pixelwiseregression_amd import kernels as K
    B = 2
    x, w, bias = rnd(B, Cin, H, H, seed=1), rnd(J, Cin, 3, 3, seed=2, scale=0.05), rnd(J, seed=3)
    ref = F.conv2d(q(x, dtype), q(w, dtype), bias.float().double(), padding=1)
    pack = K.pack_conv(w.float().to(DEV), 0, K.BF16 if dtype == torch.bfloat16 else K.F32)
    _, yn = K.conv_fwd(nhwc(x, dtype), pack, J, 3, 1, bias=bias.float().to(DEV), nhwc_out=False, nchw_out=True)
    assert_close(yn.double().cpu(), ref, fp32_out_tol(dtype), "conv nchw out")


@pytest.mark.parametrize("J,B,H,W", [(14, 4, 64, 64), (21, 3, 36, 96), (32, 9, 8, 64)])
def test_narrow_conv_pair_equals_two_single_launches(J, B, H, W):
    """pwr_conv_fwd_nchw_pair: the two heads' last convs of a stage in one launch of the narrow weight-stationary kernel -- the bytes of
    two pwr_conv_fwd launches; a shape the kernel does not take is reported, not computed some other way."""
    from pixelwiseregression_amd import kernels as K
    xa, xb = nhwc(rnd(B, 128, H, W, seed=1), torch.bfloat16), nhwc(rnd(B, 128, H, W, seed=2), torch.bfloat16)
    wa, wb = rnd(J, 128, 3, 3, seed=3, scale=0.05).float().to(DEV), rnd(J, 128, 3, 3, seed=4, scale=0.05).float().to(DEV)
    ba, bb = rnd(J, seed=5).float().to(DEV), rnd(J, seed=6).float().to(DEV)
    g = torch.Generator().manual_seed(7)
    sta = torch.stack([torch.randn(B, 128, generator=g) * 0.2, torch.ones(B, 128), torch.rand(B, 128, generator=g) + 0.5, torch.randn(B, 128, generator=g) * 0.2]).contiguous().to(DEV)
    stb = torch.stack([torch.randn(B, 128, generator=g) * 0.2, torch.ones(B, 128), torch.rand(B, 128, generator=g) + 0.5, torch.randn(B, 128, generator=g) * 0.2]).contiguous().to(DEV)
    pa, pb = K.pack_conv(wa, 0, K.BF16), K.pack_conv(wb, 0, K.BF16)
    pair = K.conv_fwd_nchw_pair(xa, pa, xb, pb, J, 3, bias_a=ba, bias_b=bb, norm_a=sta, norm_b=stb)
    assert pair is not None
    _, ya = K.conv_fwd(xa, pa, J, 3, 1, bias=ba, norm=sta, nhwc_out=False, nchw_out=True)
    _, yb = K.conv_fwd(xb, pb, J, 3, 1, bias=bb, norm=stb, nhwc_out=False, nchw_out=True)
    assert float(ya.abs().max()) > 0 and torch.equal(pair[0], ya) and torch.equal(pair[1], yb)
    # 64 input channels: not that kernel's shape
    x64 = nhwc(rnd(B, 64, H, W, seed=1), torch.bfloat16)
    p64 = K.pack_conv(rnd(J, 64, 3, 3, seed=3).float().to(DEV), 0, K.BF16)
    assert K.conv_fwd_nchw_pair(x64, p64, x64, p64, J, 3) is None


@pytest.mark.parametrize("J", [14, 21, 1, 32])
@pytest.mark.parametrize("B,H,W", [(2, 64, 64), (5, 36, 96), (1, 64, 32), (9, 8, 32)])
@pytest.mark.parametrize("form", ["plain", "norm"])
def test_narrow_weight_stationary_conv(J, B, H, W, form):
    """conv_wstat.hip KIND 3: the heads' last conv (128 -> J <= 32, fp32 NCHW out, model.py:64 / :113) as a persistent weight-stationary
    kernel whose four waves split the tile's rows.  Ragged tile counts per workgroup (B * H / 4 * W / 32 = 32 ... 135 tiles, not a
    multiple of the workgroup count; 18 tiles: one tile per workgroup), J odd / 1 / the full 32, with and without the norm + ReLU prologue
    and the bias, against float64."""
    from pixelwiseregression_amd import kernels as K
    x, w, bias = rnd(B, 128, H, W, seed=1), rnd(J, 128, 3, 3, seed=2, scale=0.05), rnd(J, seed=3)
    xq = q(x, torch.bfloat16)
    st = None
    if form == "norm":
        g = torch.Generator().manual_seed(7)
        mean, rstd = torch.randn(B, 128, generator=g) * 0.2, torch.rand(B, 128, generator=g) + 0.5
        gam, bet = torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g) * 0.2
        st = torch.stack([mean, rstd, rstd * gam[None], bet[None].expand(B, 128)]).float().contiguous().to(DEV)
        # the engine's operand: fmaf(x - mean, rstd * gamma, beta), ReLU, ONE rounding to bf16 (conv_wstat.hip stage_micro)
        xn = torch.relu((xq - mean.double()[:, :, None, None]) * (rstd * gam[None]).float().double()[:, :, None, None] + bet.double()[None, :, None, None])
        xq = q(xn.float(), torch.bfloat16)
    ref = F.conv2d(xq, q(w, torch.bfloat16), bias.float().double(), padding=1)
    pack = K.pack_conv(w.float().to(DEV), 0, K.BF16)
    _, yn = K.conv_fwd(nhwc(x, torch.bfloat16), pack, J, 3, 1, bias=bias.float().to(DEV), norm=st, nhwc_out=False, nchw_out=True)
    # (norm form: the fp32 fma rounds once where float64 does not, so a few operands land on the neighbouring bf16: 2^-9 of an operand)
    assert_close(yn.double().cpu(), ref, fp32_out_tol(torch.bfloat16, form == "norm"), "narrow conv %s" % form)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(2, 16, 16, 32, 32, 3), (2, 64, 64, 128, 128, 3), (3, 5, 7, 64, 128, 3),
                                  (2, 16, 16, 128, 64, 1), (1, 4, 4, 16, 16, 3), (2, 32, 32, 128, 16, 3),
                                  (3, 4, 4, 16, 16, 3), (3, 2, 2, 16, 16, 3), (3, 8, 8, 16, 16, 3), (3, 4, 4, 16, 32, 1),
                                  (2, 64, 64, 128, 64, 1), (2, 32, 32, 64, 128, 1), (2, 32, 64, 128, 32, 1),
                                  (2, 16, 16, 64, 64, 5), (2, 32, 32, 128, 128, 7), (3, 5, 7, 32, 16, 7), (1, 64, 64, 128, 16, 5)])
def test_conv_dgrad_stride1(case, dtype):
    """data gradient = pwr_conv_fwd on dy with the kind-1 (flipped, transposed) weight pack."""
    from pixelwiseregression_amd import kernels as K
    B, H, W, Cin, Cout, k = case
    w = rnd(Cout, Cin, k, k, seed=2, scale=(Cin * k * k) ** -0.5)
    dy = rnd(B, Cout, H, W, seed=7)
    x = torch.zeros(B, Cin, H, W, dtype=torch.float64, requires_grad=True)
    F.conv2d(x, q(w, dtype), None, padding=k // 2).backward(q(dy, dtype))
    pack = K.pack_conv(w.float().to(DEV), 1, K.BF16 if dtype == torch.bfloat16 else K.F32)
    dx, _ = K.conv_fwd(nhwc(dy, dtype), pack, Cin, k, 1)
    assert_close_out(nchw(dx), x.grad, dtype, False, "dgrad %s" % (case,))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("k", [3, 5, 7])
@pytest.mark.parametrize("B,H,Cin,Cout", [(2, 32, 64, 128), (1, 6, 16, 32), (3, 20, 128, 128), (2, 64, 128, 128)])
def test_conv_dgrad_stride2(dtype, B, H, Cin, Cout, k):
    """rows are processed in parity-class order (tiles of one class skip the taps it cannot reach); the small cases have
    tiles that straddle classes and samples."""
    from pixelwiseregression_amd import kernels as K
    w = rnd(Cout, Cin, k, k, seed=2, scale=(Cin * k * k) ** -0.5)
    dy = rnd(B, Cout, H // 2, H // 2, seed=7)
    x = torch.zeros(B, Cin, H, H, dtype=torch.float64, requires_grad=True)
    F.conv2d(x, q(w, dtype), None, stride=2, padding=k // 2).backward(q(dy, dtype))
    pack = K.pack_conv(w.float().to(DEV), 2, K.BF16 if dtype == torch.bfloat16 else K.F32)
    dx, _ = K.conv_fwd(nhwc(dy, dtype), pack, Cin, k, 1, mode=1)
    assert_close_out(nchw(dx), x.grad, dtype, False, "dgrad stride 2")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(2, 16, 16, 32, 32, 3, 1, 4), (2, 64, 64, 128, 128, 3, 1, 16), (3, 5, 7, 64, 64, 3, 1, 3),
                                  (2, 32, 64, 32, 64, 3, 1, 8), (3, 8, 32, 64, 64, 3, 1, 5), (2, 16, 16, 128, 64, 1, 1, 2), (2, 32, 32, 64, 128, 3, 2, 8), (1, 2, 2, 16, 16, 3, 1, 1),
                                  (2, 8, 8, 256, 32, 3, 1, 2), (2, 16, 16, 32, 32, 5, 1, 4), (2, 32, 32, 128, 128, 7, 1, 9),
                                  (3, 5, 7, 64, 16, 7, 1, 2), (2, 32, 32, 128, 128, 5, 2, 5), (2, 64, 64, 128, 16, 5, 1, 40)])
@pytest.mark.parametrize("prologue", [False, True])
def test_conv_wgrad(case, dtype, prologue):
    from pixelwiseregression_amd import kernels as K
    B, H, W, Cin, Cout, k, stride, splits = case
    x = rnd(B, Cin, H, W, seed=1)
    xin = q(x, dtype)
    st = None
    if prologue:
        xin, st = apply_nr(xin, B, Cin, dtype)
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    dy = rnd(B, Cout, Ho, Wo, seed=7)
    w = torch.zeros(Cout, Cin, k, k, dtype=torch.float64, requires_grad=True)
    F.conv2d(xin, w, None, stride=stride, padding=k // 2).backward(q(dy, dtype))
    dw = K.conv_wgrad(nhwc(x, dtype), nhwc(dy, dtype), Cout, k, stride, norm=st, relu_in=True, splits=splits)
    assert_close(dw.double().cpu(), w.grad, fp32_out_tol(dtype, prologue), "wgrad %s" % (case,))


@pytest.mark.parametrize("case", [(4, 64, 64, 128, 128, 3, 80), (2, 64, 64, 128, 128, 3, 100), (4, 32, 32, 64, 64, 3, 40),
                                  (8, 64, 64, 128, 64, 1, 512), (2, 16, 16, 64, 128, 1, 20), (2, 16, 16, 128, 16, 3, 7)])
def test_wgrad_split_k_reduce_is_repeatable_and_accumulates_exactly(case):
    """The split-K reduction sums the slabs in a fixed order (no atomics): the same launch twice gives the same bits, and
    accumulate adds exactly the same sums onto dw."""
    from pixelwiseregression_amd import kernels as K
    B, H, W, Cin, Cout, k, splits = case
    x, dy = nhwc(rnd(B, Cin, H, W, seed=1), torch.bfloat16), nhwc(rnd(B, Cout, H, W, seed=7), torch.bfloat16)
    dw = K.conv_wgrad(x, dy, Cout, k, 1, splits=splits).clone()
    assert float(dw.abs().max()) > 0
    assert torch.equal(dw, K.conv_wgrad(x, dy, Cout, k, 1, splits=splits))
    assert torch.equal(K.conv_wgrad(x, dy, Cout, k, 1, splits=splits, dw=dw.clone()), dw + dw)


@pytest.mark.parametrize("case", [(4, 64, 64, 128, 128, 80), (3, 20, 96, 64, 64, 24), (2, 32, 32, 128, 64, 8), (2, 8, 32, 64, 128, 3), (5, 64, 64, 128, 128, 37)])
def test_wgrad3_lds_dma_path(case):
    """conv_wgrad_dma.hip: the 3x3 weight gradient with both operands staged by LDS-DMA (taken when the operand needs no norm).
    (1) bit-identical to the kernel that applies a norm on the way -- the register-staged one for 128-wide output tiles, the in-LDS
    variant for 64-wide ones -- run with an IDENTITY norm state (mean 0, scale 1, beta 0, no ReLU: fmaf(x - 0, 1, 0) == x): same
    accumulation order, same split-K slabs; (2) matches F.conv2d's float64 weight gradient, image borders, ragged last split and all."""
    from pixelwiseregression_amd import kernels as K
    B, H, W, Cin, Cout, splits = case
    x, dy = rnd(B, Cin, H, W, seed=3), rnd(B, Cout, H, W, seed=4)
    xd, dyd = nhwc(x, torch.bfloat16), nhwc(dy, torch.bfloat16)
    ident = torch.zeros(4, B, Cin, device=DEV)
    ident[1:3] = 1.0                                    # [mean, rstd, scale, beta] = [0, 1, 1, 0]
    old = K.conv_wgrad(xd, dyd, Cout, 3, 1, norm=ident, relu_in=False, splits=splits)        # register-staged kernel
    new = K.conv_wgrad(xd, dyd, Cout, 3, 1, norm=None, splits=splits)                        # LDS-DMA kernel
    assert float(new.abs().max()) > 0 and torch.equal(old, new), float((old - new).abs().max())
    w = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(q(x, torch.bfloat16), w, None, padding=1).backward(q(dy, torch.bfloat16))
    assert_close(new.double().cpu(), w.grad, 1e-4, "LDS-DMA wgrad %s" % (case,))


@pytest.mark.parametrize("case", [(6, 64, 64, 64, 64, 37), (3, 20, 96, 64, 64, 24), (2, 32, 32, 128, 64, 8), (9, 32, 32, 64, 64, 5), (4, 128, 128, 64, 64, 40)])
def test_wgrad3_lds_dma_norm_in_lds(case):
    """conv_wgrad_dma.hip, NRM variant (taken for 64-wide output tiles): the operand's pending norm + ReLU applied IN LDS to the tile
    the LDS-DMA landed, one step ahead of the MFMAs.  Against F.conv2d's float64 weight gradient of the normalised, bf16-rounded
    operand -- splits that straddle a sample (the per-sample norm state changes inside the split), image borders, ragged last split --
    and bit-identical over repeated launches: the first build of this variant used a register the LDS read had not filled yet in
    about one launch of ten (pixelwiseregression_amd/codeobj_scan.py: async_lds_hazards is the static gate for that)."""
    from pixelwiseregression_amd import kernels as K
    B, H, W, Cin, Cout, splits = case
    x, dy = rnd(B, Cin, H, W, seed=13), rnd(B, Cout, H, W, seed=14)
    gamma, beta = (1 + 0.3 * rnd(Cin, seed=15)).float().to(DEV), (0.3 * rnd(Cin, seed=16)).float().to(DEV)
    xd, dyd = nhwc(x, torch.bfloat16), nhwc(dy, torch.bfloat16)
    st = K.norm_stats(xd, gamma, beta, mode=0)
    outs = [K.conv_wgrad(xd, dyd, Cout, 3, 1, norm=st, relu_in=True, splits=splits).clone() for _ in range(12)]
    for o in outs[1:]:
        assert torch.equal(outs[0], o), float((outs[0] - o).abs().max())
    xq = q(x, torch.bfloat16)
    mean, scale, shift = (st[i].double().cpu()[:, :, None, None] for i in (0, 2, 3))
    xin = q(torch.relu((xq - mean) * scale + shift), torch.bfloat16)
    w = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(xin, w, None, padding=1).backward(q(dy, torch.bfloat16))
    assert_close(outs[0].double().cpu(), w.grad, 2e-3, "LDS-DMA wgrad with the norm in LDS %s" % (case,))


@pytest.mark.parametrize("case", [(2, 128, 128, 128, 128, 16), (3, 64, 64, 64, 128, 7), (2, 24, 64, 128, 128, 5), (5, 64, 128, 128, 128, 37)])
@pytest.mark.parametrize("prologue", [False, True])
def test_wgrad3_stride2(case, prologue):
    """conv_wgrad3_kernel<..., STR = 2>: the weight gradient of the stride-2 3x3 conv (model.py:182, the stem's last layer) on the three-tap
    kernel -- 65 staged input pixels per 32-pixel K tile, fragments from every second staged row.  Against F.conv2d's float64 weight gradient
    (image borders: input row / column -1 are the conv's zero padding; splits that straddle samples and leave a ragged last split), and
    repeatable bit for bit."""
    from pixelwiseregression_amd import kernels as K
    B, H, W, Cin, Cout, splits = case
    x = rnd(B, Cin, H, W, seed=21)
    xin = q(x, torch.bfloat16)
    st = None
    if prologue:
        xin, st = apply_nr(xin, B, Cin, torch.bfloat16)
    dy = rnd(B, Cout, H // 2, W // 2, seed=22)
    w = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(xin, w, None, stride=2, padding=1).backward(q(dy, torch.bfloat16))
    xd, dyd = nhwc(x, torch.bfloat16), nhwc(dy, torch.bfloat16)
    dw = K.conv_wgrad(xd, dyd, Cout, 3, 2, norm=st, relu_in=True, splits=splits).clone()
    assert_close(dw.double().cpu(), w.grad, 2e-3 if prologue else 1e-4, "stride-2 three-tap wgrad %s" % (case,))
    for _ in range(3):
        assert torch.equal(dw, K.conv_wgrad(xd, dyd, Cout, 3, 2, norm=st, relu_in=True, splits=splits))


@pytest.mark.parametrize("case", [(2, 64, 64, 128, 16, 14, 7), (3, 8, 128, 32, 64, 64, 5), (2, 64, 64, 64, 64, 64, 9), (2, 16, 64, 64, 128, 128, 3),
                                  (5, 64, 64, 128, 32, 21, 80), (3, 12, 192, 64, 64, 64, 11),
                                  # odd steps_per_split with many splits: the 64-pixel re-cut leaves EMPTY trailing splits (exact-size tensors)
                                  (18, 64, 64, 128, 16, 14, 256), (6, 64, 64, 64, 64, 64, 256)])
@pytest.mark.parametrize("prologue", [False, True])
def test_wgrad3_64_pixel_steps(case, prologue):
    """conv_wgrad3_kernel<..., KPX = 64>: the narrow three-tap layers (<= 32 output channels: the heads' 128 -> J conv; 64-channel tiles) on
    maps whose width is a multiple of 64 take K steps of 64 pixels.  Against F.conv2d's float64 weight gradient (zero-padded output
    channels, splits with an odd number of 32-pixel steps -- the 64-pixel steps then cut the K range elsewhere --, splits that straddle
    samples), repeatable bit for bit."""
    from pixelwiseregression_amd import kernels as K
    B, H, W, Cin, Cout, cr, splits = case
    x, dy = rnd(B, Cin, H, W, seed=31), rnd(B, Cout, H, W, seed=32)
    dy[:, cr:] = 0
    xin = q(x, torch.bfloat16)
    st = None
    if prologue:
        xin, st = apply_nr(xin, B, Cin, torch.bfloat16)
    w = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(xin, w, None, padding=1).backward(q(dy, torch.bfloat16))
    xd, dyd = nhwc(x, torch.bfloat16), nhwc(dy, torch.bfloat16)
    dw = K.conv_wgrad(xd, dyd, cr, 3, 1, norm=st, relu_in=True, splits=splits).clone()
    assert_close(dw.double().cpu(), w.grad[:cr], 2e-3 if prologue else 1e-4, "three-tap wgrad, 64-pixel steps %s" % (case,))
    for _ in range(3):
        assert torch.equal(dw, K.conv_wgrad(xd, dyd, cr, 3, 1, norm=st, relu_in=True, splits=splits))


def test_grouped_weight_gradients_match_float64():
    """pwr_conv_wgrad_group: the 24 conv layers of one stage's small-map ResBlocks (16x16 .. 2x2, 1x1 128->64, 3x3 64->64, 1x1 64->128,
    norm + ReLU on the operand load) in ONE grouped launch, each against F.conv2d's float64 weight gradient; repeatable bit for bit."""
    from pixelwiseregression_amd import kernels as K
    B = 6
    jobs, refs = [], []
    seed = 0
    for W in (16, 16, 8, 8, 4, 4, 2, 2):
        for (cin, cout, k) in ((128, 64, 1), (64, 64, 3), (64, 128, 1)):
            seed += 1
            x, dy = rnd(B, cin, W, W, seed=seed), rnd(B, cout, W, W, seed=100 + seed)
            gamma, beta = (1 + 0.2 * rnd(cin, seed=200 + seed)).float().to(DEV), (0.2 * rnd(cin, seed=300 + seed)).float().to(DEV)
            xd = nhwc(x, torch.bfloat16)
            st = K.norm_stats(xd, gamma, beta, mode=0)
            xq = q(x, torch.bfloat16)
            mean, scale, shift = (st[i].double().cpu()[:, :, None, None] for i in (0, 2, 3))
            xin = q(torch.relu((xq - mean) * scale + shift), torch.bfloat16)
            w = torch.zeros(cout, cin, k, k, dtype=torch.float64, requires_grad=True)
            F.conv2d(xin, w, None, padding=k // 2).backward(q(dy, torch.bfloat16))
            jobs.append((xd, nhwc(dy, torch.bfloat16), k, st))
            refs.append(w.grad)
    dws = K.conv_wgrad_group(jobs)
    for i, (dw, ref) in enumerate(zip(dws, refs)):
        assert_close(dw.double().cpu(), ref, 2e-3, "grouped wgrad job %d %s" % (i, tuple(ref.shape)))
    again = K.conv_wgrad_group(jobs)
    assert all(torch.equal(a, b_) for a, b_ in zip(dws, again))


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_wgrad_padded_dy(dtype):
    """head's last conv: dy arrives as [B,P,P,Jp] with Jp > J zero-padded channels"""
    from pixelwiseregression_amd import kernels as K
    B, P, Cin, J, Jp = 2, 16, 64, 14, 16
    x, g = rnd(B, Cin, P, P, seed=1), rnd(B, J, P, P, seed=2)
    w = torch.zeros(J, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(q(x, dtype), w, None, padding=1).backward(q(g, dtype))
    dyp = K.nchw_to_nhwc_pad(g.float().to(DEV), Jp, dtype)
    assert dyp.shape == (B, P, P, Jp) and float(dyp[..., J:].abs().max()) == 0.0
    assert_close(dyp[..., :J].double().cpu().permute(0, 3, 1, 2), q(g.float().double(), dtype), 1e-7, "transpose")
    dw = K.conv_wgrad(nhwc(x, dtype), dyp, J, 3, 1, splits=4)
    assert_close(dw.double().cpu(), w.grad, fp32_out_tol(dtype), "wgrad padded")
    # and the matching data gradient: K dimension = Jp with a pack built from the J real channels
    wt = rnd(J, Cin, 3, 3, seed=9, scale=0.05)
    xg = torch.zeros(B, Cin, P, P, dtype=torch.float64, requires_grad=True)
    F.conv2d(xg, q(wt, dtype), None, padding=1).backward(q(g, dtype))
    pack = K.pack_conv(wt.float().to(DEV), 1, K.BF16 if dtype == torch.bfloat16 else K.F32)
    dx, _ = K.conv_fwd(dyp, pack, Cin, 3, 1)
    assert_close_out(nchw(dx), xg.grad, dtype, False, "dgrad padded")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("S", [32, 128])
@pytest.mark.parametrize("k", [3, 5, 7])
def test_stem_conv(dtype, S, k):
    from pixelwiseregression_amd import kernels as K
    B, C0 = 3, 32
    img, w, b = rnd(B, 1, S, S, seed=1), rnd(C0, 1, k, k, seed=2), rnd(C0, seed=3)
    ref = F.conv2d(img.float().double(), w.float().double(), b.float().double(), padding=k // 2)
    y = K.stem_conv_fwd(img.float().to(DEV), w.float().to(DEV), b.float().to(DEV), dtype)
    assert_close(nchw(y), ref, 1e-6 if dtype == torch.float32 else 5e-3, "stem fwd")
    dy = rnd(B, C0, S, S, seed=4)
    wz = torch.zeros(C0, 1, k, k, dtype=torch.float64, requires_grad=True)
    F.conv2d(img.float().double(), wz, None, padding=k // 2).backward(q(dy, dtype))
    dw = K.stem_conv_wgrad(img.float().to(DEV), nhwc(dy, dtype), k)
    assert_close(dw.double().cpu(), wz.grad, 2e-5, "stem wgrad")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("J,P,Fo", [(14, 64, 128), (4, 16, 32), (21, 10, 64)])
def test_catconv(dtype, J, P, Fo):
    from pixelwiseregression_amd import kernels as K
    B = 2
    pm, dm, lb = rnd(B, J, P, P, seed=1).abs(), rnd(B, J, P, P, seed=2), rnd(B, 1, P, P, seed=3)
    w, b = rnd(Fo, 2 * J + 1, 1, 1, seed=4, scale=0.2), rnd(Fo, seed=5)
    cat = torch.cat([pm, dm, lb], 1).float().double().requires_grad_()
    wd = w.float().double().requires_grad_()
    bd = b.float().double().requires_grad_()
    ref = F.conv2d(cat, wd, bd)
    y = K.catconv_fwd(pm.float().to(DEV), dm.float().to(DEV), lb.float().to(DEV), w.float().view(Fo, -1).contiguous().to(DEV),
                      b.float().to(DEV), dtype)
    assert_close(nchw(y), ref.detach(), 1e-5 if dtype == torch.float32 else 5e-3, "catconv fwd")
    dy = rnd(B, Fo, P, P, seed=6)
    ref.backward(q(dy, dtype))
    gp, gd = K.catconv_dgrad(nhwc(dy, dtype), w.float().view(Fo, -1).contiguous().to(DEV), J)
    assert_close(gp.double().cpu(), cat.grad[:, :J], 2e-5, "catconv dgrad p")
    assert_close(gd.double().cpu(), cat.grad[:, J:2 * J], 2e-5, "catconv dgrad d")
    dw, db = K.catconv_wgrad(pm.float().to(DEV), dm.float().to(DEV), lb.float().to(DEV), nhwc(dy, dtype))
    assert_close(dw.double().cpu(), wd.grad, 2e-5, "catconv wgrad")
    assert_close(db.double().cpu(), bd.grad, 2e-5, "catconv bgrad")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,C", [(2, 64, 64, 128), (3, 2, 2, 64), (1, 128, 128, 32), (2, 5, 7, 16), (32, 4, 4, 64),
                                     (3, 4, 4, 16), (3, 2, 2, 16), (3, 8, 8, 32), (3, 16, 16, 16)])
def test_instance_norm_stats_and_backward(dtype, B, H, W, C):
    from pixelwiseregression_amd import kernels as K
    y = rnd(B, C, H, W, seed=1) * (1 + rnd(1, C, 1, 1, seed=2).abs()) + 3 * rnd(1, C, 1, 1, seed=3)   # offsets >> std
    gamma, beta = 1 + 0.2 * rnd(C, seed=4), 0.3 * rnd(C, seed=5)
    yq = q(y, dtype).requires_grad_()
    gd, bd = gamma.float().double().requires_grad_(), beta.float().double().requires_grad_()
    out = torch.relu(F.instance_norm(yq, None, None, gd, bd, True, 0.1, 1e-5))
    yd = nhwc(y, dtype)
    state = K.norm_stats(yd, gamma.float().to(DEV), beta.float().to(DEV), mode=0)
    mean, rstd, scale, shift = state[0], state[1], state[2], state[3]
    mref = yq.detach().mean(dim=(2, 3))
    vref = yq.detach().var(dim=(2, 3), unbiased=False)
    assert_close(mean.double().cpu(), mref, 1e-5, "mean")
    assert_close(rstd.double().cpu(), (vref + 1e-5).rsqrt(), 2e-5, "rstd")
    # applying scale/shift reproduces relu(norm(y))
    app = torch.relu((yq.detach() - mean.double().cpu()[:, :, None, None]) * scale.double().cpu()[:, :, None, None]
                     + shift.double().cpu()[:, :, None, None])
    assert_close(app, out.detach(), 2e-5, "apply")
    g = rnd(B, C, H, W, seed=6)
    add = rnd(B, C, H, W, seed=7)
    out.backward(q(g, dtype))
    dy, dgam, dbet = K.norm_bwd(nhwc(g, dtype), yd, state, relu=True, addend=nhwc(add, dtype))
    t = 5e-5 if dtype == torch.float32 else 2e-2
    assert_close(nchw(dy), yq.grad + q(add, dtype), t, "norm bwd dy")
    assert_close(dgam.double().cpu(), gd.grad, t, "dgamma")
    assert_close(dbet.double().cpu(), bd.grad, t, "dbeta")


@pytest.mark.parametrize("dtype", DTYPES)
def test_batch_norm_train_and_eval(dtype):
    from pixelwiseregression_amd import kernels as K
    B, H, W, C = 4, 16, 16, 64
    y = rnd(B, C, H, W, seed=1) + 2 * rnd(1, C, 1, 1, seed=3)
    gamma, beta = 1 + 0.2 * rnd(C, seed=4), 0.3 * rnd(C, seed=5)
    rm, rv = 0.1 * rnd(C, seed=8), 1 + 0.1 * rnd(C, seed=9).abs()
    yq = q(y, dtype).requires_grad_()
    gd, bd = gamma.float().double().requires_grad_(), beta.float().double().requires_grad_()
    rm_ref, rv_ref = rm.float().double().clone(), rv.float().double().clone()
    out = torch.relu(F.batch_norm(yq, rm_ref, rv_ref, gd, bd, True, 0.1, 1e-5))
    yd = nhwc(y, dtype)
    rmd, rvd = rm.float().to(DEV), rv.float().to(DEV)
    state = K.norm_stats(yd, gamma.float().to(DEV), beta.float().to(DEV), mode=1, running_mean=rmd, running_var=rvd)
    mean, scale, shift = state[0], state[2], state[3]
    assert_close(rmd.double().cpu(), rm_ref, 1e-5, "running_mean")
    assert_close(rvd.double().cpu(), rv_ref, 1e-5, "running_var")
    app = torch.relu((yq.detach() - mean.double().cpu()[:, :, None, None]) * scale.double().cpu()[:, :, None, None]
                     + shift.double().cpu()[:, :, None, None])
    assert_close(app, out.detach(), 2e-5, "bn apply")
    g = rnd(B, C, H, W, seed=6)
    out.backward(q(g, dtype))
    dy, dgam, dbet = K.norm_bwd(nhwc(g, dtype), yd, state, relu=True, mode=1)
    t = 5e-5 if dtype == torch.float32 else 2e-2
    assert_close(nchw(dy), yq.grad, t, "bn bwd dy")
    assert_close(dgam.double().cpu(), gd.grad, t, "bn dgamma")
    # eval mode
    ref_eval = torch.relu(F.batch_norm(q(y, dtype), rm.float().double(), rv.float().double(), gamma.float().double(),
                                       beta.float().double(), False, 0.1, 1e-5))
    st2 = K.norm_stats(yd, gamma.float().to(DEV), beta.float().to(DEV), mode=2, running_mean=rm.float().to(DEV),
                       running_var=rv.float().to(DEV)).double().cpu()
    app = torch.relu((q(y, dtype) - st2[0][:, :, None, None]) * st2[2][:, :, None, None] + st2[3][:, :, None, None])
    assert_close(app, ref_eval, 2e-5, "bn eval apply")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,C", [(2, 64, 64, 128), (2, 5, 7, 16), (3, 2, 2, 64)])
def test_maxpool_and_upsample(dtype, B, H, W, C):
    from pixelwiseregression_amd import kernels as K
    x = q(rnd(B, C, H, W, seed=1), dtype).requires_grad_()
    h = F.max_pool2d(x, 2, stride=2)
    xd = nhwc(x.detach(), dtype)
    hd = K.maxpool_fwd(xd)
    assert_close(nchw(hd), h.detach(), 1e-7, "maxpool fwd")
    h2 = q(rnd(*h.shape, seed=2), dtype).requires_grad_()
    up = F.interpolate(h2, size=(H, W)) + x
    upd = K.upsample_add(nhwc(h2.detach(), dtype), xd)
    assert_close(nchw(upd), q(up.detach(), dtype), 1e-7 if dtype == torch.float32 else 8e-3, "upsample add")
    g = q(rnd(B, C, H, W, seed=3), dtype)
    gh = q(rnd(*h.shape, seed=4), dtype)
    (up * g).sum().backward()
    dh = K.upsample_bwd(nhwc(g, dtype), h.shape[2], h.shape[3])
    assert_close(nchw(dh), h2.grad, 1e-6 if dtype == torch.float32 else 8e-3, "upsample bwd")
    x.grad = None
    (h * gh).sum().backward()
    dx = K.maxpool_bwd(xd, nhwc(gh, dtype), addend=nhwc(g, dtype))
    assert_close(nchw(dx), q(x.grad + g, dtype), 1e-6 if dtype == torch.float32 else 8e-3, "maxpool bwd")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("J,P", [(14, 64), (4, 16), (21, 10), (42, 32)])
def test_concat_nhwc_roundtrip_and_stage_input_conv(dtype, J, P):
    """model.py:208 concat as NHWC + model.py:137 1x1 conv on the MFMA path, fwd / wgrad / dgrad, vs torch on the CPU."""
    from pixelwiseregression_amd import kernels as K
    B, Fo = 2, 64
    pm, dm, lb = rnd(B, J, P, P, seed=1).abs(), rnd(B, J, P, P, seed=2), rnd(B, 1, P, P, seed=3)
    cat = torch.cat([pm, dm, lb], 1).float().double()
    xc = K.cat_to_nhwc(pm.float().to(DEV), dm.float().to(DEV), lb.float().to(DEV), dtype)
    Cp = xc.shape[-1]
    assert Cp % 8 == 0 and Cp >= 2 * J + 1
    assert_close(nchw(xc)[:, :2 * J + 1], q(cat, dtype), 1e-7, "concat")
    assert float(xc[..., 2 * J + 1:].abs().max()) == 0.0 if Cp > 2 * J + 1 else True
    w = rnd(Fo, 2 * J + 1, 1, 1, seed=4, scale=0.2)
    catq = q(cat, dtype).requires_grad_()
    wd = q(w, dtype).requires_grad_()
    ref = F.conv2d(catq, wd, None)
    dt = K.BF16 if dtype == torch.bfloat16 else K.F32
    y, _ = K.conv_fwd(xc, K.pack_conv(w.float().to(DEV), 0, dt), Fo, 1, 1)
    assert_close_out(nchw(y), ref.detach(), dtype, False, "stage-in fwd")
    dy = rnd(B, Fo, P, P, seed=6)
    ref.backward(q(dy, dtype))
    dw = K.conv_wgrad(xc, nhwc(dy, dtype), Fo, 1, 1, splits=4, cin_real=2 * J + 1)
    assert dw.shape == (Fo, 2 * J + 1, 1, 1)
    assert_close(dw.double().cpu(), wd.grad, fp32_out_tol(dtype), "stage-in wgrad")
    dx, _ = K.conv_fwd(nhwc(dy, dtype), K.pack_conv(w.float().to(DEV), 1, dt), Cp, 1, 1)
    gp, gd = K.nhwc_to_cat_grad(dx, J)
    # (fp32 outputs of nhwc_to_cat_grad, converted from the data gradient the conv stored in the activation dtype: one bf16 rounding per
    # element on top of the fp32-output bound)
    for got_, ref_, nm in ((gp, catq.grad[:, :J], "p"), (gd, catq.grad[:, J:2 * J], "d")):
        if dtype == torch.float32:
            assert_close(got_.double().cpu(), ref_, 2e-5, "stage-in dgrad " + nm)
        else:
            assert_close_out(got_.double().cpu(), ref_, dtype, False, "stage-in dgrad " + nm)


@pytest.mark.parametrize("dtype", DTYPES)
def test_bias_gradient_sums(dtype):
    from pixelwiseregression_amd import kernels as K
    x = rnd(3, 64, 17, 9, seed=1)
    assert_close(K.colsum_nhwc(nhwc(x, dtype)).double().cpu(), q(x, dtype).sum(dim=(0, 2, 3)), 1e-5, "colsum")
    x = rnd(2, 128, 64, 64, seed=2)
    assert_close(K.colsum_nhwc(nhwc(x, dtype)).double().cpu(), q(x, dtype).sum(dim=(0, 2, 3)), 1e-5, "colsum big")
    g = rnd(5, 14, 64, 64, seed=3).float()
    assert_close(K.planesum_nchw(g.to(DEV)).double().cpu(), g.double().sum(dim=(0, 2, 3)), 1e-5, "planesum")


def test_norm_fused_handoff_stress():
    """The single-launch InstanceNorm statistics hand partial sums from all blocks of a sample to the last-arriving block
    (agent-scope release/acquire).  Alternate two inputs through the SAME workspace 300 times: a stale or torn read of a
    partial would reproduce the other input's statistics.  Every result must equal the first (validated) one bit for bit."""
    from pixelwiseregression_amd import kernels as K, _lib
    import torch
    B, H, W, C = 32, 64, 64, 128
    gamma, beta = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    xs = [(torch.randn(B, H, W, C, device=DEV) * (1 + i) + 3 * i).to(torch.bfloat16) for i in range(2)]
    l = _lib.lib()
    partial = torch.zeros(l.pwr_norm_partial_bytes(B, H * W, C) // 4, dtype=torch.float32, device=DEV)
    state = torch.empty(4, B, C, dtype=torch.float32, device=DEV)

    def run(x):
        _lib.check(l.pwr_norm_stats(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), None, None, partial.data_ptr(), state.data_ptr(),
                                    B, H * W, C, 0, 1e-5, 0.1, K.BF16, _lib.stream_ptr(x.device)), "pwr_norm_stats")
        return state.clone()

    refs = []
    for x in xs:
        st = run(x)
        m = x.float().mean(dim=(1, 2))
        assert (st[0] - m).abs().max().item() < 1e-3 * max(1.0, m.abs().max().item())
        refs.append(st)
    # a small kernel in between keeps other CUs' caches warm with lines of the workspace
    for it in range(300):
        i = it & 1
        st = run(xs[i])
        assert torch.equal(st, refs[i]), "iteration %d: hand-off returned different statistics" % it


# ---------------------------------------------------------------- fused small-map ResBlock (model.py:6-23)
@pytest.mark.parametrize("B,H", [(3, 2), (2, 4), (5, 8), (2, 16), (33, 4)])
def test_resblock_small_fused_vs_unfused(B, H):
    """One-launch ResBlock == the unfused kernel sequence (same rounding points), forward and backward, and both close to a
    float64 torch ResBlock evaluated on the bf16-rounded parameters."""
    from pixelwiseregression_amd import kernels as K
    C, Fh, dt = 128, 64, torch.bfloat16
    assert K.resblock_small_supported(H, H, C, 0, K.BF16)
    x = rnd(B, C, H, H, seed=1)
    ws = [rnd(Fh, C, 1, 1, seed=2, scale=C ** -0.5), rnd(Fh, Fh, 3, 3, seed=3, scale=(9 * Fh) ** -0.5), rnd(C, Fh, 1, 1, seed=4, scale=Fh ** -0.5)]
    bs = [rnd(Fh, seed=5, scale=0.1), rnd(Fh, seed=6, scale=0.1), rnd(C, seed=7, scale=0.1)]
    gs = [1 + 0.2 * rnd(C, seed=8), 1 + 0.2 * rnd(Fh, seed=9), 1 + 0.2 * rnd(Fh, seed=10)]
    bes = [0.2 * rnd(C, seed=11), 0.2 * rnd(Fh, seed=12), 0.2 * rnd(Fh, seed=13)]
    gout = rnd(B, C, H, H, seed=14)
    dev = lambda t: t.float().to(DEV)
    xd, gd_ = nhwc(x, dt), nhwc(gout, dt)
    wf = [K.pack_conv(dev(w), 0, K.BF16) for w in ws]
    wd = [K.pack_conv(dev(w), 1, K.BF16) for w in ws]
    bd, gmd, bed = [dev(t) for t in bs], [dev(t) for t in gs], [dev(t) for t in bes]

    # ---- unfused sequence
    sa = K.norm_stats(xd, gmd[0], bed[0])
    t1, _ = K.conv_fwd(xd, wf[0], Fh, 1, 1, bias=bd[0], norm=sa)
    sb = K.norm_stats(t1, gmd[1], bed[1])
    t2, _ = K.conv_fwd(t1, wf[1], Fh, 3, 1, bias=bd[1], norm=sb)
    sc = K.norm_stats(t2, gmd[2], bed[2])
    out, _ = K.conv_fwd(t2, wf[2], C, 1, 1, bias=bd[2], norm=sc, residual=xd)
    g2, _ = K.conv_fwd(gd_, wd[2], Fh, 1, 1)
    dt2, dgc, dbc = K.norm_bwd(g2, t2, sc)
    g1, _ = K.conv_fwd(dt2, wd[1], Fh, 3, 1)
    dt1, dgb, dbb = K.norm_bwd(g1, t1, sb)
    g0, _ = K.conv_fwd(dt1, wd[0], C, 1, 1)
    dx, dga, dba = K.norm_bwd(g0, xd, sa, addend=gd_)

    # ---- fused
    fout, ft1, ft2, fst = K.resblock_fwd_small(xd, wf, bd, gmd, bed)
    # (backward on the UNFUSED forward's tensors and states: identical ReLU masks, so only summation order differs)
    fdx, fdt1, fdt2, fsums, fpg = K.resblock_bwd_small(gd_, xd, t1, t2, wd, [sa, sb, sc])
    torch.cuda.synchronize()

    def close(a, b, rel, what):
        assert_close(a.double().cpu(), b.double().cpu(), rel, what)
    # same rounding points: differences are single bf16 ulps where an fp32 sum order tipped a rounding
    for a, b, what in ((fst[0], sa, "state a"), (fst[1], sb, "state b"), (fst[2], sc, "state c")):
        close(a, b, 2e-2, what)
    close(ft1, t1, 1e-2, "t1"); close(ft2, t2, 2e-2, "t2"); close(fout, out, 2e-2, "out")
    close(fdt2, dt2, 3e-2, "dt2"); close(fdt1, dt1, 3e-2, "dt1"); close(fdx, dx, 3e-2, "dx")
    for s_, dg, db, what in ((fsums[0], dga, dba, "a"), (fsums[1], dgb, dbb, "b"), (fsums[2], dgc, dbc, "c")):
        close(s_[:, 1].sum(0), dg, 3e-2, "dgamma " + what)
        close(s_[:, 0].sum(0), db, 3e-2, "dbeta " + what)
    for got, ref_, what in zip(fpg[:6], (dga, dba, dgb, dbb, dgc, dbc), ("dga", "dba", "dgb", "dbb", "dgc", "dbc")):
        close(got, ref_, 3e-2, "param grads kernel " + what)
    close(fpg[6], K.colsum_nhwc(gd_), 1e-3, "bias gradient of conv c")

    # ---- float64 torch ResBlock on the bf16-rounded operands
    xq = q(x, dt).requires_grad_(True)
    wq = [q(w, dt) for w in ws]
    h = F.conv2d(torch.relu(F.instance_norm(xq, weight=gs[0].float().double(), bias=bes[0].float().double(), eps=1e-5)), wq[0], bs[0].float().double())
    h = F.conv2d(torch.relu(F.instance_norm(h, weight=gs[1].float().double(), bias=bes[1].float().double(), eps=1e-5)), wq[1], bs[1].float().double(), padding=1)
    h = F.conv2d(torch.relu(F.instance_norm(h, weight=gs[2].float().double(), bias=bes[2].float().double(), eps=1e-5)), wq[2], bs[2].float().double())
    ref = xq + h
    ref.backward(q(gout, dt))
    assert_close(nchw(fout), ref.detach(), 3e-2, "fused out vs float64")
    if H > 2:   # (4-pixel instance norm is too ill-conditioned for a bf16 gradient comparison)
        # bf16 roundings of t1 / t2 flip ReLUs of near-zero pre-activations: an L2 criterion, not a max-norm one
        rel_l2 = ((nchw(fdx) - xq.grad).norm() / xq.grad.norm()).item()
        assert rel_l2 < 5e-2, "fused dx vs float64: rel L2 %.3e" % rel_l2


# ---------------------------------------------------------------- conv epilogue column statistics
STATS_CASES = [
    # B, H, W, Cin, Cout, k   (patch kernel: W % 32 == 0; universal: H*W % 128 == 0)
    (2, 64, 64, 128, 128, 3), (3, 32, 32, 64, 64, 3), (2, 8, 32, 32, 64, 3), (2, 32, 32, 128, 64, 1), (3, 16, 16, 64, 128, 1),
    (2, 64, 64, 16, 128, 3), (1, 128, 128, 32, 64, 3), (2, 64, 64, 64, 128, 1), (2, 32, 64, 128, 32, 1),
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", STATS_CASES)
def test_conv_epilogue_forward_stats(case, dtype):
    """conv + statistics from its epilogue + finalize == conv, then the standalone norm statistics."""
    from pixelwiseregression_amd import kernels as K
    B, H, W, Cin, Cout, k = case
    kd = K.BF16 if dtype == torch.bfloat16 else K.F32
    assert K.conv_stats_chunks(H, W, Cin, Cout, k, 1, 0, kd) > 0
    x = nhwc(rnd(B, Cin, H, W, seed=1), dtype)
    w = rnd(Cout, Cin, k, k, seed=2, scale=(Cin * k * k) ** -0.5)
    bias = (rnd(Cout, seed=3) * 20).float().to(DEV)         # means far from 0: the sums must be shifted
    gamma, beta = (1 + 0.2 * rnd(Cout, seed=4)).float().to(DEV), (0.2 * rnd(Cout, seed=5)).float().to(DEV)
    res = nhwc(rnd(B, Cout, H, W, seed=6), dtype)
    pack = K.pack_conv(w.float().to(DEV), 0, kd)
    for mode in (0, 1):
        y0, _ = K.conv_fwd(x, pack, Cout, k, 1, bias=bias, residual=res)
        ref = K.norm_stats(y0, gamma, beta, mode=mode)
        y1, partial, chunks = K.conv_fwd_stats(x, pack, Cout, k, 1, bias=bias, residual=res)
        assert torch.equal(y0, y1)
        assert not torch.isnan(partial).any()
        st = K.norm_finalize_partial(partial, chunks, gamma, beta, B, H * W, mode=mode)
        # mean / rstd / scale agree to fp32 summation noise (both are statistics of the same stored tensor)
        assert_close(st.double().cpu(), ref.double().cpu(), 2e-5 if dtype == torch.float32 else 2e-4, "state (mode %d)" % mode)


@pytest.mark.parametrize("form", ["plain", "norm"])
@pytest.mark.parametrize("case", [(2, 64, 64, 128), (3, 36, 96, 128), (2, 64, 64, 64), (3, 36, 96, 64), (1, 128, 128, 64), (5, 8, 64, 64)])
def test_weight_stationary_conv_statistics_against_the_standalone_ones(case, form):
    """The shapes csrc/conv_wstat.hip takes (3x3, 128 / 64 -> 128 channels, no residual) with the forward statistics from its epilogue:
    the conv equals the plain launch bit for bit, and the statistics -- finalised -- equal the standalone norm statistics of the stored
    tensor.  The 64-channel form sums its column groups in another order and takes an fp32 shift (the order of the kernel it replaced):
    both orders have to give the same statistics to summation noise."""
    from pixelwiseregression_amd import kernels as K
    B, H, W, Cin = case
    x = nhwc(rnd(B, Cin, H, W, seed=1), torch.bfloat16)
    w = rnd(128, Cin, 3, 3, seed=2, scale=(Cin * 9) ** -0.5)
    bias = (rnd(128, seed=3) * 20).float().to(DEV)          # means far from 0: the sums must be shifted
    gamma, beta = (1 + 0.2 * rnd(128, seed=4)).float().to(DEV), (0.2 * rnd(128, seed=5)).float().to(DEV)
    st = K.norm_stats(x, torch.ones(Cin, device=DEV), torch.zeros(Cin, device=DEV), mode=0) if form == "norm" else None
    pack = K.pack_conv(w.float().to(DEV), 0, K.BF16)
    y0, _ = K.conv_fwd(x, pack, 128, 3, 1, bias=bias, norm=st)
    y1, partial, chunks = K.conv_fwd_stats(x, pack, 128, 3, 1, bias=bias, norm=st)
    assert float(y0.float().abs().max()) > 0 and torch.equal(y0, y1) and not torch.isnan(partial).any()
    for mode in (0, 1):
        ref = K.norm_stats(y0, gamma, beta, mode=mode)
        got = K.norm_finalize_partial(partial, chunks, gamma, beta, B, H * W, mode=mode)
        assert_close(got.double().cpu(), ref.double().cpu(), 2e-4, "state (mode %d)" % mode)
    # (round 6) two finalisations of one shape in one launch -- the two heads' norms of one depth -- equal the single launches bit for bit
    one_a = K.norm_finalize_partial(partial, chunks, gamma, beta, B, H * W, mode=0)
    one_b = K.norm_finalize_partial(partial, chunks, beta, gamma, B, H * W, mode=0)
    two_a, two_b = K.norm_finalize_partial_pair(partial, gamma, beta, partial, beta, gamma, chunks, B, H * W)
    assert torch.equal(one_a, two_a) and torch.equal(one_b, two_b)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", STATS_CASES)
def test_conv_epilogue_norm_backward_sums(case, dtype):
    """data-gradient conv + norm-backward sums from its epilogue == data gradient, then the three-launch norm backward."""
    from pixelwiseregression_amd import kernels as K
    B, H, W, Cin, Cout, k = case
    kd = K.BF16 if dtype == torch.bfloat16 else K.F32
    dyn = nhwc(rnd(B, Cin, H, W, seed=1), dtype)             # gradient w.r.t. the conv that FOLLOWS the norm
    w = rnd(Cin, Cout, k, k, seed=2, scale=(Cin * k * k) ** -0.5)   # that conv maps Cout -> Cin channels
    pack_d = K.pack_conv(w.float().to(DEV), 1, kd)
    y = nhwc(rnd(B, Cout, H, W, seed=3), dtype)             # pre-norm tensor of the norm being back-propagated
    gamma, beta = (1 + 0.2 * rnd(Cout, seed=4)).float().to(DEV), (0.2 * rnd(Cout, seed=5)).float().to(DEV)
    add = nhwc(rnd(B, Cout, H, W, seed=6), dtype)
    for mode in (0, 1):
        state = K.norm_stats(y, gamma, beta, mode=mode)
        g0, _ = K.conv_fwd(dyn, pack_d, Cout, k, 1)
        dy0, dg0, db0 = K.norm_bwd(g0, y, state, addend=add, mode=mode)
        g1, partial, chunks = K.conv_fwd_stats(dyn, pack_d, Cout, k, 1, nb_y=y, nb_state=state)
        assert torch.equal(g0, g1)
        assert not torch.isnan(partial).any()
        dy1, dg1, db1 = K.norm_bwd_from_partial(g1, y, state, partial, chunks, addend=add, mode=mode)
        if mode == 0:      # (round 6) the one-launch form of the same step: bit-identical, with the skip addend too
            dy2, dg2, db2 = K.norm_bwd_fold(g1, y, state, partial, chunks, addend=add)
            assert torch.equal(dy1, dy2) and torch.equal(dg1, dg2) and torch.equal(db1, db2)
        t = 2e-5 if dtype == torch.float32 else 1e-2
        assert_close(dy1.double().cpu(), dy0.double().cpu(), t, "dy (mode %d)" % mode)
        assert_close(dg1.double().cpu(), dg0.double().cpu(), 1e-4, "dgamma")
        assert_close(db1.double().cpu(), db0.double().cpu(), 1e-4, "dbeta")


@pytest.mark.parametrize("B,H,W,C", [(2, 32, 32, 128), (3, 8, 64, 64), (5, 64, 64, 128)])
def test_norm_backward_pair_launch(B, H, W, C):
    """pwr_norm_bwd_from_partial_pair: the norm backwards of two tensors of one shape (the two heads' norms of one depth) as two launches
    instead of four, bit-identical to two single calls."""
    from pixelwiseregression_amd import kernels as K
    dt = torch.bfloat16
    jobs = []
    for sd in (0, 50):
        dyn = nhwc(rnd(B, C, H, W, seed=1 + sd), dt)
        w = rnd(C, C, 3, 3, seed=2 + sd, scale=(C * 9) ** -0.5)
        pack_d = K.pack_conv(w.float().to(DEV), 1, K.BF16)
        y = nhwc(rnd(B, C, H, W, seed=3 + sd), dt)
        gamma, beta = (1 + 0.2 * rnd(C, seed=4 + sd)).float().to(DEV), (0.2 * rnd(C, seed=5 + sd)).float().to(DEV)
        state = K.norm_stats(y, gamma, beta, mode=0)
        g, partial, chunks = K.conv_fwd_stats(dyn, pack_d, C, 3, 1, nb_y=y, nb_state=state)
        jobs.append((g, y, state, partial, chunks))
    singles = [K.norm_bwd_from_partial(g, y, st, pa, ch, mode=0) for (g, y, st, pa, ch) in jobs]
    (ga, ya, sa, pa, ch), (gb, yb, sb, pb, _) = jobs
    pair = K.norm_bwd_from_partial_pair(ga, ya, sa, pa, gb, yb, sb, pb, ch)
    for one, two in zip(singles, pair):
        for a, b_ in zip(one, two):
            assert torch.equal(a, b_)
    # round 6: the reduction folded into the apply launch (every workgroup sums the slab rows of its sample itself, in the reduction launch's
    # order) + the parameter sums as a launch of their own: the same bits, singly and as a pair
    fold_pair = K.norm_bwd_fold(ga, ya, sa, pa, ch, pair=(gb, yb, sb, pb))
    fold_single = [K.norm_bwd_fold(g, y, st, pt, c_) for (g, y, st, pt, c_) in jobs]
    for one, two, three in zip(singles, fold_pair, fold_single):
        for a, b_, c_ in zip(one, two, three):
            assert float(a.float().abs().max()) > 0 and torch.equal(a, b_) and torch.equal(a, c_)
    # ... and the parameter sums of several layers in one launch (what the engine issues once per backward segment)
    grouped = K.norm_bwd_params_group([(pt, c_, H * W, C) for (_, _, _, pt, c_) in jobs] * 21, B)      # (42 jobs: two launches of <= 40)
    for k_, (dg, db) in enumerate(grouped):
        assert torch.equal(dg, singles[k_ % 2][1]) and torch.equal(db, singles[k_ % 2][2])


@pytest.mark.parametrize("B,H,W", [(2, 32, 32), (3, 8, 64), (2, 64, 64)])
def test_stride2_dgrad_norm_backward_sums(B, H, W):
    """The data gradient of the stride-2 3x3 conv (four parity classes in one launch) + the norm-backward sums of the tensor it produces from
    the class launches' epilogues (one slab, 4 x tiles rows per sample) == the same data gradient, then the three-launch norm backward."""
    from pixelwiseregression_amd import kernels as K
    C, dtype = 128, torch.bfloat16
    dyn = nhwc(rnd(B, C, H, W, seed=1), dtype)                      # gradient of the stride-2 conv's output (H x W)
    w = rnd(C, C, 3, 3, seed=2, scale=(C * 9) ** -0.5)
    pack_d = K.pack_conv(w.float().to(DEV), 2, K.BF16)
    y = nhwc(rnd(B, C, 2 * H, 2 * W, seed=3), dtype)                # pre-norm tensor the conv read (2H x 2W)
    gamma, beta = (1 + 0.2 * rnd(C, seed=4)).float().to(DEV), (0.2 * rnd(C, seed=5)).float().to(DEV)
    state = K.norm_stats(y, gamma, beta, mode=0)
    g0, _ = K.conv_fwd(dyn, pack_d, C, 3, 1, mode=1)
    dy0, dg0, db0 = K.norm_bwd(g0, y, state, mode=0)
    g1, partial, chunks = K.conv_fwd_stats(dyn, pack_d, C, 3, 1, nb_y=y, nb_state=state, mode=1)
    assert chunks == 4 * (H // 4) * (W // 32)
    assert torch.equal(g0, g1)
    assert not torch.isnan(partial).any()
    dy1, dg1, db1 = K.norm_bwd_from_partial(g1, y, state, partial, chunks, mode=0)
    assert_close(dy1.double().cpu(), dy0.double().cpu(), 1e-2, "dy")
    assert_close(dg1.double().cpu(), dg0.double().cpu(), 1e-4, "dgamma")
    assert_close(db1.double().cpu(), db0.double().cpu(), 1e-4, "dbeta")
    g2, partial2, _ = K.conv_fwd_stats(dyn, pack_d, C, 3, 1, nb_y=y, nb_state=state, mode=1)
    assert torch.equal(partial, partial2) and torch.equal(g1, g2)


@pytest.mark.parametrize("B,H", [(3, 2), (2, 4), (5, 8), (2, 16), (33, 4)])
@pytest.mark.parametrize("which", ["up", "pool", "both"])
def test_resblock_small_bwd_fused_neighbours(B, H, which):
    """pwr_resblock_bwd_small_x: the one-launch ResBlock backward summing the up-sample's gradient while it loads (model.py:45) and / or
    routing the max-pool's gradient while it stores (model.py:40).  Everything it writes -- g_out, dx, dt1, dt2, the norm sums, the bias
    sums, the pooled tensor's input gradient -- is BIT-identical to pwr_upsample_bwd, pwr_resblock_bwd_small, pwr_maxpool_bwd in sequence."""
    from pixelwiseregression_amd import kernels as K
    C, Fh, dt = 128, 64, torch.bfloat16
    ws = [rnd(Fh, C, 1, 1, seed=2, scale=C ** -0.5), rnd(Fh, Fh, 3, 3, seed=3, scale=(9 * Fh) ** -0.5), rnd(C, Fh, 1, 1, seed=4, scale=Fh ** -0.5)]
    bs = [rnd(Fh, seed=5, scale=0.1), rnd(Fh, seed=6, scale=0.1), rnd(C, seed=7, scale=0.1)]
    gs = [1 + 0.2 * rnd(C, seed=8), 1 + 0.2 * rnd(Fh, seed=9), 1 + 0.2 * rnd(Fh, seed=10)]
    bes = [0.2 * rnd(C, seed=11), 0.2 * rnd(Fh, seed=12), 0.2 * rnd(Fh, seed=13)]
    dev = lambda t: t.float().to(DEV)
    wf = [K.pack_conv(dev(w), 0, K.BF16) for w in ws]
    wd = [K.pack_conv(dev(w), 1, K.BF16) for w in ws]
    bd, gmd, bed = [dev(t) for t in bs], [dev(t) for t in gs], [dev(t) for t in bes]
    a = nhwc(rnd(B, C, 2 * H, 2 * H, seed=21), dt)           # the level above: the tensor that was pooled
    a[:, ::2, ::2] = a[:, 1::2, ::2]                          # (ties inside windows: the FIRST maximum in scan order takes the gradient)
    x = K.maxpool_fwd(a) if which != "up" else nhwc(rnd(B, C, H, H, seed=22), dt)
    up = nhwc(rnd(B, C, 2 * H, 2 * H, seed=23), dt)          # gradient of the level above's `out`
    addend = nhwc(rnd(B, C, 2 * H, 2 * H, seed=24), dt)      # gradient already sitting on the pooled tensor's skip path
    gout_plain = nhwc(rnd(B, C, H, H, seed=25), dt)
    _, t1, t2, st = K.resblock_fwd_small(x, wf, bd, gmd, bed)
    # ---- separate launches
    gout0 = K.upsample_bwd(up, H, H) if which != "pool" else gout_plain
    dx0, dt10, dt20, sums0, _ = K.resblock_bwd_small(gout0, x, t1, t2, wd, st)
    pd0 = K.maxpool_bwd(a, dx0, addend) if which != "up" else None
    # ---- one launch
    dx1, dt11, dt21, sums1, gout1, pd1, _ = K.resblock_bwd_small_x(x, t1, t2, wd, st, gout=None if which != "pool" else gout_plain,
                                                                  up_src=up if which != "pool" else None,
                                                                  pool_a=a if which != "up" else None, pool_addend=addend if which != "up" else None)
    assert torch.equal(gout1, gout0)
    assert torch.equal(dx1, dx0) and torch.equal(dt11, dt10) and torch.equal(dt21, dt20)
    for s0, s1 in zip(sums0, sums1):
        assert torch.equal(s0, s1)
    if which != "up":
        assert not torch.isnan(pd1.float()).any() and torch.equal(pd1, pd0)


@pytest.mark.parametrize("B,H,xmode", [(3, 2, 1), (2, 4, 1), (5, 8, 1), (2, 16, 1), (33, 4, 1), (2, 4, 2), (5, 8, 2), (3, 16, 2)])
def test_resblock_small_fused_input(B, H, xmode):
    """pwr_resblock_fwd_small_x: the one-launch ResBlock computing its input on the fly -- xmode 1: x = maxpool2x2(a) (model.py:40),
    xmode 2: x = nearest-upsample(h) + a (model.py:45-47) -- and writing x where the backward pass expects it.  Everything it writes
    (x, t1, t2, out, the three norm states) is BIT-identical to the stand-alone pool / up-sample kernel followed by
    pwr_resblock_fwd_small."""
    from pixelwiseregression_amd import kernels as K
    C, Fh, dt = 128, 64, torch.bfloat16
    ws = [rnd(Fh, C, 1, 1, seed=2, scale=C ** -0.5), rnd(Fh, Fh, 3, 3, seed=3, scale=(9 * Fh) ** -0.5), rnd(C, Fh, 1, 1, seed=4, scale=Fh ** -0.5)]
    bs = [rnd(Fh, seed=5, scale=0.1), rnd(Fh, seed=6, scale=0.1), rnd(C, seed=7, scale=0.1)]
    gs = [1 + 0.2 * rnd(C, seed=8), 1 + 0.2 * rnd(Fh, seed=9), 1 + 0.2 * rnd(Fh, seed=10)]
    bes = [0.2 * rnd(C, seed=11), 0.2 * rnd(Fh, seed=12), 0.2 * rnd(Fh, seed=13)]
    dev = lambda t: t.float().to(DEV)
    wf = [K.pack_conv(dev(w), 0, K.BF16) for w in ws]
    bd, gmd, bed = [dev(t) for t in bs], [dev(t) for t in gs], [dev(t) for t in bes]
    if xmode == 1:
        a = nhwc(rnd(B, C, 2 * H, 2 * H, seed=21), dt)
        h = None
        x_ref = K.maxpool_fwd(a)
    else:
        a = nhwc(rnd(B, C, H, H, seed=21), dt)
        h = nhwc(rnd(B, C, H // 2, H // 2, seed=22), dt)
        x_ref = K.upsample_add(h, a)
    out0, t10, t20, st0 = K.resblock_fwd_small(x_ref, wf, bd, gmd, bed)
    x1, out1, t11, t21, st1 = K.resblock_fwd_small_x(xmode, a, h, wf, bd, gmd, bed)
    assert torch.equal(x1, x_ref)
    assert torch.equal(out1, out0) and torch.equal(t11, t10) and torch.equal(t21, t20)
    for s0, s1 in zip(st0, st1):
        assert torch.equal(s0, s1)


@pytest.mark.parametrize("B,H,W", [(3, 64, 64), (2, 8, 32), (33, 64, 64)])
def test_conv_pair_launch(B, H, W):
    """pwr_conv_fwd_stats_pair: the two heads' convs of one depth (3x3 128 -> 128, different inputs, weights, biases and norm states) in
    ONE launch -- outputs and statistics rows bit-identical to the two single launches; one job with a norm prologue, one without
    (the heads' first convs read the raw hourglass output)."""
    from pixelwiseregression_amd import kernels as K
    C = 128
    xa, xb = nhwc(rnd(B, C, H, W, seed=31), torch.bfloat16), nhwc(rnd(B, C, H, W, seed=32), torch.bfloat16)
    wa = K.pack_conv(rnd(C, C, 3, 3, seed=33, scale=(9 * C) ** -0.5).float().to(DEV), 0, K.BF16)
    wb = K.pack_conv(rnd(C, C, 3, 3, seed=34, scale=(9 * C) ** -0.5).float().to(DEV), 0, K.BF16)
    ba, bb = rnd(C, seed=35, scale=0.1).float().to(DEV), rnd(C, seed=36, scale=0.1).float().to(DEV)
    sta = K.norm_stats(xa, (1 + 0.2 * rnd(C, seed=37)).float().to(DEV), (0.2 * rnd(C, seed=38)).float().to(DEV))
    for na, nb in ((sta, None), (None, None)):
        ya, pa, _ = K.conv_fwd_stats(xa, wa, C, 3, 1, bias=ba, norm=na)
        yb, pb, _ = K.conv_fwd_stats(xb, wb, C, 3, 1, bias=bb, norm=nb)
        (ya2, pa2), (yb2, pb2), _ = K.conv_fwd_stats_pair(xa, wa, xb, wb, C, 3, bias_a=ba, bias_b=bb, norm_a=na, norm_b=nb)
        assert torch.equal(ya, ya2) and torch.equal(yb, yb2)
        assert torch.equal(pa, pa2) and torch.equal(pb, pb2)


@pytest.mark.parametrize("B,H,W,Cin", [(3, 64, 64, 128), (2, 8, 32, 32), (5, 64, 64, 32), (2, 16, 64, 64)])
def test_conv_dgrad_pair_launch(B, H, W, Cin):
    """pwr_conv_dgrad_stats_pair: the two heads' data gradients of one depth (kind-1 packs, norm-backward sums in the epilogue; 128 <- 128
    for the middle convs, 128 <- 32 / 64 for the last conv's padded J channels) in ONE launch: gradients and slab rows bit-identical to
    the two single launches."""
    from pixelwiseregression_amd import kernels as K
    C = 128
    dya, dyb = nhwc(rnd(B, Cin, H, W, seed=41), torch.bfloat16), nhwc(rnd(B, Cin, H, W, seed=42), torch.bfloat16)
    wa = K.pack_conv(rnd(Cin, C, 3, 3, seed=43, scale=(9 * C) ** -0.5).float().to(DEV), 1, K.BF16)
    wb = K.pack_conv(rnd(Cin, C, 3, 3, seed=44, scale=(9 * C) ** -0.5).float().to(DEV), 1, K.BF16)
    ya, yb = nhwc(rnd(B, C, H, W, seed=45), torch.bfloat16), nhwc(rnd(B, C, H, W, seed=46), torch.bfloat16)
    sta = K.norm_stats(ya, (1 + 0.2 * rnd(C, seed=47)).float().to(DEV), (0.2 * rnd(C, seed=48)).float().to(DEV))
    stb = K.norm_stats(yb, (1 + 0.2 * rnd(C, seed=49)).float().to(DEV), (0.2 * rnd(C, seed=50)).float().to(DEV))
    xa, pa, _ = K.conv_fwd_stats(dya, wa, C, 3, 1, relu_in=False, nb_y=ya, nb_state=sta)
    xb, pb, _ = K.conv_fwd_stats(dyb, wb, C, 3, 1, relu_in=False, nb_y=yb, nb_state=stb)
    (xa2, pa2), (xb2, pb2), _ = K.conv_dgrad_stats_pair(dya, wa, ya, sta, dyb, wb, yb, stb, C, 3)
    assert float(xa.float().abs().max()) > 0
    assert torch.equal(xa, xa2) and torch.equal(xb, xb2)
    assert torch.equal(pa, pa2) and torch.equal(pb, pb2)


@pytest.mark.parametrize("B,H,W", [(3, 64, 64), (2, 8, 32), (2, 16, 96)])
def test_conv_dgrad_pair_with_the_norm_backward_folded_in(B, H, W):
    """pwr_conv_dgrad_fold_stats_pair: two layers of the heads' backward chain.  Reference: data gradient of the layer above (raw g + slab) ->
    pwr_norm_bwd_apply_from_partial (dy) -> pwr_conv_dgrad_stats_pair on dy.  Folded: the second launch reads the RAW g and the slab and
    applies the norm backward in its staging -- data gradients, their slab rows and the dy it writes out equal the reference bit for bit
    (image borders, several tiles per sample, more than one sample)."""
    from pixelwiseregression_amd import kernels as K
    C, bf = 128, torch.bfloat16
    top = [nhwc(rnd(B, C, H, W, seed=61 + i), bf) for i in range(2)]                       # dy of the layer above (already folded)
    w_up = [K.pack_conv(rnd(C, C, 3, 3, seed=63 + i, scale=(9 * C) ** -0.5).float().to(DEV), 1, K.BF16) for i in range(2)]
    w_lo = [K.pack_conv(rnd(C, C, 3, 3, seed=65 + i, scale=(9 * C) ** -0.5).float().to(DEV), 1, K.BF16) for i in range(2)]
    y_mid = [nhwc(rnd(B, C, H, W, seed=67 + i), bf) for i in range(2)]                     # pre-norm tensors between the two layers
    y_low = [nhwc(rnd(B, C, H, W, seed=69 + i), bf) for i in range(2)]                     # ... and below the lower layer
    st_mid = [K.norm_stats(y_mid[i], (1 + 0.2 * rnd(C, seed=71 + i)).float().to(DEV), (0.2 * rnd(C, seed=73 + i)).float().to(DEV)) for i in range(2)]
    st_low = [K.norm_stats(y_low[i], (1 + 0.2 * rnd(C, seed=75 + i)).float().to(DEV), (0.2 * rnd(C, seed=77 + i)).float().to(DEV)) for i in range(2)]
    (ga, pa), (gb, pb), chunks = K.conv_dgrad_stats_pair(top[0], w_up[0], y_mid[0], st_mid[0], top[1], w_up[1], y_mid[1], st_mid[1], C, 3)
    dya = K.norm_bwd_fold(ga.clone(), y_mid[0], st_mid[0], pa, chunks)
    dyb = K.norm_bwd_fold(gb.clone(), y_mid[1], st_mid[1], pb, chunks)
    dya, dyb = (d[0] if isinstance(d, (tuple, list)) else d for d in (dya, dyb))
    (xa, qa), (xb, qb), _ = K.conv_dgrad_stats_pair(dya, w_lo[0], y_low[0], st_low[0], dyb, w_lo[1], y_low[1], st_low[1], C, 3)
    (xa2, qa2, da2), (xb2, qb2, db2), _ = K.conv_dgrad_fold_stats_pair(ga, w_lo[0], y_low[0], st_low[0], y_mid[0], st_mid[0], pa,
                                                                      gb, w_lo[1], y_low[1], st_low[1], y_mid[1], st_mid[1], pb, chunks, C, 3)
    assert float(xa.float().abs().max()) > 0 and not torch.isnan(da2.float()).any()
    assert torch.equal(da2, dya) and torch.equal(db2, dyb)
    assert torch.equal(xa2, xa) and torch.equal(xb2, xb)
    assert torch.equal(qa2, qa) and torch.equal(qb2, qb)


@pytest.mark.parametrize("case", [(4, 64, 64, 128, 128, 85), (5, 64, 64, 128, 128, 37), (3, 20, 96, 128, 128, 24), (2, 8, 32, 128, 256, 3), (2, 32, 32, 256, 128, 8),
                                  (32, 64, 64, 128, 128, 85), (2, 128, 128, 64, 128, 40), (3, 20, 96, 64, 128, 12), (5, 8, 32, 64, 256, 3)])
@pytest.mark.parametrize("prologue", [False, True])
def test_wgrad3_wave_specialised(case, prologue):
    """conv_wgrad_ws.hip (whole 128-channel tiles: 4 MFMA waves + 4 loader waves per workgroup, the operand's norm + ReLU applied in LDS by
    the loader waves): against F.conv2d's float64 weight gradient of the normalised, bf16-rounded operand -- splits that straddle a sample
    (the per-sample norm state changes inside a split), image borders, a ragged last split, several channel tiles -- and bit-identical over
    repeated launches; with an identity norm state the in-LDS pass gives the bits of the no-norm form."""
    from pixelwiseregression_amd import kernels as K
    B, H, W, Cin, Cout, splits = case
    x, dy = rnd(B, Cin, H, W, seed=23), rnd(B, Cout, H, W, seed=24)
    xd, dyd = nhwc(x, torch.bfloat16), nhwc(dy, torch.bfloat16)
    st = None
    xin = q(x, torch.bfloat16)
    if prologue:
        gamma, beta = (1 + 0.3 * rnd(Cin, seed=25)).float().to(DEV), (0.3 * rnd(Cin, seed=26)).float().to(DEV)
        st = K.norm_stats(xd, gamma, beta, mode=0)
        mean, scale, shift = (st[i].double().cpu()[:, :, None, None] for i in (0, 2, 3))
        xin = q(torch.relu((xin - mean) * scale + shift), torch.bfloat16)
    outs = [K.conv_wgrad(xd, dyd, Cout, 3, 1, norm=st, relu_in=True, splits=splits).clone() for _ in range(6)]
    for o in outs[1:]:
        assert torch.equal(outs[0], o), float((outs[0] - o).abs().max())
    if B <= 8:
        w = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
        F.conv2d(xin, w, None, padding=1).backward(q(dy, torch.bfloat16))
        assert_close(outs[0].double().cpu(), w.grad, 2e-3 if prologue else 1e-4, "wave-specialised wgrad %s" % (case,))
    if not prologue:
        ident = torch.zeros(4, B, Cin, device=DEV)
        ident[1:3] = 1.0                                    # [mean, rstd, scale, beta] = [0, 1, 1, 0]
        assert torch.equal(outs[0], K.conv_wgrad(xd, dyd, Cout, 3, 1, norm=ident, relu_in=False, splits=splits))


@pytest.mark.parametrize("case", [(4, 64, 64, 128, 128, 42), (3, 20, 96, 128, 128, 12), (2, 8, 32, 256, 128, 2)])
@pytest.mark.parametrize("prologue", [False, True])
def test_wgrad_pair_launch(case, prologue):
    """pwr_conv_wgrad_pair: the two heads' weight gradients of one depth in ONE launch + ONE reduce: bit-identical to two single calls with
    the same split count; without a norm both jobs may read the same x (the heads' first convs both read the hourglass output)."""
    from pixelwiseregression_amd import kernels as K
    B, H, W, Cin, Cout, splits = case
    xa, xb = nhwc(rnd(B, Cin, H, W, seed=51), torch.bfloat16), nhwc(rnd(B, Cin, H, W, seed=52), torch.bfloat16)
    dya, dyb = nhwc(rnd(B, Cout, H, W, seed=53), torch.bfloat16), nhwc(rnd(B, Cout, H, W, seed=54), torch.bfloat16)
    sta = stb = None
    if prologue:
        sta = K.norm_stats(xa, (1 + 0.3 * rnd(Cin, seed=55)).float().to(DEV), (0.3 * rnd(Cin, seed=56)).float().to(DEV))
        stb = K.norm_stats(xb, (1 + 0.3 * rnd(Cin, seed=57)).float().to(DEV), (0.3 * rnd(Cin, seed=58)).float().to(DEV))
    else:
        xb = xa
    da = K.conv_wgrad(xa, dya, Cout, 3, 1, norm=sta, splits=splits)
    db = K.conv_wgrad(xb, dyb, Cout, 3, 1, norm=stb, splits=splits)
    pa, pb = K.conv_wgrad_pair(xa, dya, xb, dyb, norm_a=sta, norm_b=stb, splits=splits)
    assert float(da.abs().max()) > 0 and torch.equal(da, pa) and torch.equal(db, pb)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,C", [(2, 64, 64, 128), (3, 8, 32, 64), (1, 5, 7, 32)])
def test_norm_apply_and_the_weight_gradient_from_the_materialised_operand(dtype, B, H, W, C):
    """pwr_norm_apply (round 6): relu(norm(y)) written out as a tensor equals fmaf(y - mean, scale, beta), ReLU, one rounding, in float64
    arithmetic up to that rounding; and the weight gradient fed the materialised operand without a norm is BIT-IDENTICAL to the weight
    gradient that applies the norm on load (what the engine relies on for the heads' norm-fed layers)."""
    from pixelwiseregression_amd import kernels as K
    y = rnd(B, C, H, W, seed=21)
    yq, st = q(y, dtype), None
    ref, st = apply_nr(yq, B, C, dtype)
    got = K.norm_apply(nhwc(y, dtype), st)
    if dtype == torch.float32:
        assert_close(nchw(got), ref, 2e-6, "norm_apply")
    else:
        assert_close_out(nchw(got), ref, dtype, True, "norm_apply")
    if W % 32 == 0 and C == 128:
        dy = nhwc(rnd(B, C, H, W, seed=22), dtype)
        a = K.conv_wgrad(nhwc(y, dtype), dy, C, 3, 1, norm=st, splits=8)
        b_ = K.conv_wgrad(got, dy, C, 3, 1, norm=None, splits=8)
        assert float(a.abs().max()) > 0 and torch.equal(a, b_)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,C", [(32, 32, 32, 128), (3, 64, 64, 128), (2, 32, 64, 64), (5, 24, 40, 32), (2, 16, 16, 128)])
def test_norm_statistics_with_the_producer_fused_in(dtype, B, H, W, C):
    """pwr_norm_stats_fused_src (round 6): the max-pool / the up-sample + skip add computed inside the statistics launch of the InstanceNorm that
    follows -- the tensor AND the [4][B][C] state are the bytes of the separate launches (pwr_maxpool_fwd / pwr_upsample_add_fwd, then
    pwr_norm_stats); H W <= 512 has no fused form (the engine then issues the separate launches)."""
    from pixelwiseregression_amd import kernels as K
    gamma, beta = (1 + 0.2 * rnd(C, seed=4)).float().to(DEV), (0.2 * rnd(C, seed=5)).float().to(DEV)
    big = nhwc(rnd(B, C, 2 * H, 2 * W, seed=31) + 3.0, dtype)
    skip, low = nhwc(rnd(B, C, H, W, seed=32) - 2.0, dtype), nhwc(rnd(B, C, H // 2, W // 2, seed=33), dtype)
    for src, (xa, xh), sep in ((1, (big, None), lambda: K.maxpool_fwd(big)), (2, (skip, low), lambda: K.upsample_add(low, skip))):
        y0 = sep()
        st0 = K.norm_stats(y0, gamma, beta, mode=0)
        got = K.norm_stats_fused_src(src, xa, xh, gamma, beta)
        if H * W <= 512:
            assert got is None
            continue
        y1, st1 = got
        assert float(y0.float().abs().max()) > 0 and torch.equal(y0, y1) and torch.equal(st0, st1), (src, float((st0 - st1).abs().max()))
