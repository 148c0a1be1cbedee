"""Known-answer cases: every bf16 kernel family of the library (and a few fp32 ones) on FIXED inputs, reduced to a sha256 per case.

Why: round 4 met ONE lease out of eleven on which the final library produced other bf16 results (19 engine-level failures, cause never
found, the list of failing kernels lost).  Every reduction in the library has a fixed order (no float atomics), so on a healthy box the
bytes a kernel writes for given input bytes are a constant of the build.  `tests/golden/kat_digests.json` holds those constants;
`tests/test_00_kat_gpu.py` (first file of the suite) compares and NAMES the cases that differ, so that a disagreeing box says which
kernel family disagrees before anything else runs; `tools/kat.py --write` regenerates the file after a change of arithmetic.

The inputs come from an integer hash (splitmix64 of the element index), not from a library RNG: the same bytes on every machine,
numpy / torch version and device.
"""
import hashlib

import numpy as np
import torch

DEV = "cuda:0"
BF = torch.bfloat16


def det(shape, seed, scale=1.0, offset=0.0):
    """float32 tensor of `shape`, uniform in [-scale, scale) + offset, a pure function of (index, seed)."""
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        z = np.arange(n, dtype=np.uint64) + np.uint64(seed + 1) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    return torch.from_numpy(((2.0 * u - 1.0) * scale + offset).astype(np.float32).reshape(shape))


def dev(shape, seed, scale=1.0, offset=0.0, dtype=torch.float32):
    return det(shape, seed, scale, offset).to(dtype).to(DEV)


def digest(tensors):
    h = hashlib.sha256()
    for t in tensors:
        t = t.detach().contiguous()
        if t.dtype == torch.bfloat16:
            t = t.view(torch.int16)
        h.update(t.cpu().numpy().tobytes())
    return h.hexdigest()


def _state(B, C, seed):
    """a norm state [4,B,C] = mean, rstd, scale, beta"""
    st = torch.stack([det((B, C), seed, 0.5), torch.ones(B, C), det((B, C), seed + 1, 0.2, 1.0), det((B, C), seed + 2, 0.3)])
    return st.contiguous().to(DEV)


def _pack(K, cout, cin, k, kind, seed, dtype=1):
    w = det((cout, cin, k, k), seed, (cin * k * k) ** -0.5 * 1.7)
    return K.pack_conv(w.to(DEV), kind, dtype)


def _conv(B, H, W, Cin, Cout, k, stride=1, prologue=True, residual=False, seed=0, dtype=BF):
    def run():
        from pixelwiseregression_amd import kernels as K
        x = dev((B, H, W, Cin), seed, dtype=dtype)
        pack = _pack(K, Cout, Cin, k, 0, seed + 1, K.BF16 if dtype == BF else K.F32)
        bias = dev((Cout,), seed + 2, 0.1)
        st = _state(B, Cin, seed + 3) if prologue else None
        Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
        res = dev((B, Ho, Wo, Cout), seed + 6, dtype=dtype) if residual else None
        y, _ = K.conv_fwd(x, pack, Cout, k, stride, bias=bias, norm=st, relu_in=True, residual=res)
        return [y]
    return run


def _conv_nchw(B, H, Cin, J, seed):
    def run():
        from pixelwiseregression_amd import kernels as K
        x = dev((B, H, H, Cin), seed, dtype=BF)
        pack = _pack(K, J, Cin, 3, 0, seed + 1)
        _, yn = K.conv_fwd(x, pack, J, 3, 1, bias=dev((J,), seed + 2, 0.1), norm=_state(B, Cin, seed + 3), nhwc_out=False, nchw_out=True)
        return [yn]
    return run


def _conv_stats(B, H, W, Cin, Cout, k, seed, stride=1):
    def run():
        from pixelwiseregression_amd import kernels as K
        x = dev((B, H, W, Cin), seed, dtype=BF)
        pack = _pack(K, Cout, Cin, k, 0, seed + 1)
        y, partial, _ = K.conv_fwd_stats(x, pack, Cout, k, stride, bias=dev((Cout,), seed + 2, 2.0), norm=_state(B, Cin, seed + 3))
        return [y, partial]
    return run


def _dgrad_stats(B, H, W, Cin, Cout, k, seed, mode=0):
    def run():
        from pixelwiseregression_amd import kernels as K
        dy = dev((B, H, W, Cin), seed, dtype=BF)
        w = det((Cin, Cout, k, k), seed + 1, (Cin * k * k) ** -0.5 * 1.7)
        pack = K.pack_conv(w.to(DEV), 2 if mode else 1, K.BF16)
        s = 2 if mode else 1
        y = dev((B, s * H, s * W, Cout), seed + 2, dtype=BF)
        g, partial, _ = K.conv_fwd_stats(dy, pack, Cout, k, 1, nb_y=y, nb_state=_state(B, Cout, seed + 3), mode=mode)
        return [g, partial]
    return run


def _pair_fwd(B, H, Cin, seed):
    def run():
        from pixelwiseregression_amd import kernels as K
        xa, xb = dev((B, H, H, Cin), seed, dtype=BF), dev((B, H, H, Cin), seed + 10, dtype=BF)
        wa, wb = _pack(K, 128, Cin, 3, 0, seed + 1), _pack(K, 128, Cin, 3, 0, seed + 11)
        (ya, pa), (yb, pb), _ = K.conv_fwd_stats_pair(xa, wa, xb, wb, 128, 3, bias_a=dev((128,), seed + 2, 2.0), bias_b=dev((128,), seed + 12, 2.0),
                                                      norm_a=_state(B, Cin, seed + 3), norm_b=_state(B, Cin, seed + 13))
        return [ya, pa, yb, pb]
    return run


def _pair_dgrad(B, H, Cin, seed):
    def run():
        from pixelwiseregression_amd import kernels as K
        da, db = dev((B, H, H, Cin), seed, dtype=BF), dev((B, H, H, Cin), seed + 10, dtype=BF)
        wa = K.pack_conv(det((Cin, 128, 3, 3), seed + 1, 0.05).to(DEV), 1, K.BF16)
        wb = K.pack_conv(det((Cin, 128, 3, 3), seed + 11, 0.05).to(DEV), 1, K.BF16)
        ya, yb = dev((B, H, H, 128), seed + 2, dtype=BF), dev((B, H, H, 128), seed + 12, dtype=BF)
        (xa, pa), (xb, pb), _ = K.conv_dgrad_stats_pair(da, wa, ya, _state(B, 128, seed + 3), db, wb, yb, _state(B, 128, seed + 13), 128, 3)
        return [xa, pa, xb, pb]
    return run


def _wgrad(B, H, W, Cin, Cout, k, stride, splits, prologue, seed, cout_real=None, dtype=BF):
    def run():
        from pixelwiseregression_amd import kernels as K
        x = dev((B, H, W, Cin), seed, dtype=dtype)
        Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
        dy = dev((B, Ho, Wo, Cout), seed + 1, dtype=dtype)
        cr = cout_real or Cout
        if cr < Cout:
            dy[..., cr:] = 0
        st = _state(B, Cin, seed + 3) if prologue else None
        return [K.conv_wgrad(x, dy, cr, k, stride, norm=st, relu_in=True, splits=splits)]
    return run


def _wgrad_pair(B, seed):
    def run():
        from pixelwiseregression_amd import kernels as K
        xa, xb = dev((B, 64, 64, 128), seed, dtype=BF), dev((B, 64, 64, 128), seed + 10, dtype=BF)
        da, db = dev((B, 64, 64, 128), seed + 1, dtype=BF), dev((B, 64, 64, 128), seed + 11, dtype=BF)
        return list(K.conv_wgrad_pair(xa, da, xb, db, norm_a=_state(B, 128, seed + 3), norm_b=_state(B, 128, seed + 13), splits=24))
    return run


def _wgrad_group(B, seed):
    def run():
        from pixelwiseregression_amd import kernels as K
        jobs = []
        s = seed
        for W in (16, 8, 4, 2):
            for (cin, cout, k) in ((128, 64, 1), (64, 64, 3), (64, 128, 1)):
                s += 5
                jobs.append((dev((B, W, W, cin), s, dtype=BF), dev((B, W, W, cout), s + 1, dtype=BF), k, _state(B, cin, s + 2)))
        return K.conv_wgrad_group(jobs)
    return run


def _resblock(B, H, seed):
    def run():
        from pixelwiseregression_amd import kernels as K
        C, Fh = 128, 64
        x, gout = dev((B, H, H, C), seed, dtype=BF), dev((B, H, H, C), seed + 1, dtype=BF)
        shapes = [(Fh, C, 1), (Fh, Fh, 3), (C, Fh, 1)]
        ws = [det((co, ci, k, k), seed + 2 + i, (ci * k * k) ** -0.5 * 1.7).to(DEV) for i, (co, ci, k) in enumerate(shapes)]
        wf, wd = [K.pack_conv(w, 0, K.BF16) for w in ws], [K.pack_conv(w, 1, K.BF16) for w in ws]
        bd = [dev((c,), seed + 5 + i, 0.1) for i, c in enumerate((Fh, Fh, C))]
        gm = [dev((c,), seed + 8 + i, 0.2, 1.0) for i, c in enumerate((C, Fh, Fh))]
        be = [dev((c,), seed + 11 + i, 0.2) for i, c in enumerate((C, Fh, Fh))]
        out, t1, t2, st = K.resblock_fwd_small(x, wf, bd, gm, be)
        dx, dt1, dt2, sums, pg = K.resblock_bwd_small(gout, x, t1, t2, wd, st)
        return [out, t1, t2] + list(st) + [dx, dt1, dt2] + list(sums) + list(pg)
    return run


def _norm(B, H, W, C, seed, dtype=BF):
    def run():
        from pixelwiseregression_amd import kernels as K
        y = dev((B, H, W, C), seed, 1.0, 0.7, dtype=dtype)
        g, add = dev((B, H, W, C), seed + 1, dtype=dtype), dev((B, H, W, C), seed + 2, dtype=dtype)
        st = K.norm_stats(y, dev((C,), seed + 3, 0.2, 1.0), dev((C,), seed + 4, 0.3), mode=0)
        dy, dg, db = K.norm_bwd(g, y, st, relu=True, addend=add)
        return [st, dy, dg, db]
    return run


def _norm_from_partial(B, H, C, seed):
    def run():
        from pixelwiseregression_amd import kernels as K
        outs = []
        jobs = []
        for sd in (0, 40):
            dyn = dev((B, H, H, C), seed + sd, dtype=BF)
            pack = K.pack_conv(det((C, C, 3, 3), seed + sd + 1, 0.05).to(DEV), 1, K.BF16)
            y = dev((B, H, H, C), seed + sd + 2, 1.0, 0.5, dtype=BF)
            st = K.norm_stats(y, dev((C,), seed + sd + 3, 0.2, 1.0), dev((C,), seed + sd + 4, 0.3), mode=0)
            g, partial, chunks = K.conv_fwd_stats(dyn, pack, C, 3, 1, nb_y=y, nb_state=st)
            jobs.append((g, y, st, partial, chunks))
            outs += list(K.norm_bwd_from_partial(g, y, st, partial, chunks, mode=0))
        (ga, ya, sa, pa, ch), (gb, yb, sb, pb, _) = jobs
        for o in K.norm_bwd_from_partial_pair(ga, ya, sa, pa, gb, yb, sb, pb, ch):
            outs += list(o)
        return outs
    return run


def _finalize(B, H, C, seed):
    def run():
        from pixelwiseregression_amd import kernels as K
        x = dev((B, H, H, C), seed, dtype=BF)
        pack = _pack(K, C, C, 3, 0, seed + 1)
        y, partial, chunks = K.conv_fwd_stats(x, pack, C, 3, 1, bias=dev((C,), seed + 2, 2.0))
        return [K.norm_finalize_partial(partial, chunks, dev((C,), seed + 3, 0.2, 1.0), dev((C,), seed + 4, 0.3), B, H * H, mode=0)]
    return run


def _decoder(B, J, P, seed, method="softmax"):
    def run():
        from pixelwiseregression_amd import ops
        z, D = dev((B, J, P, P), seed, 3.0), dev((B, J, P, P), seed + 1)
        m = (det((B, 1, P, P), seed + 2) > 0.2).float().to(DEV)
        L = dev((B, 1, P, P), seed + 3) * m
        w = dev((J, 1), seed + 4, 0.2, 1.0) if method == "softmax" else None
        p, uvd = ops.decode_forward(z, D, L, m, w, method)
        gz, gD, gw = ops.decode_backward(p, z, D, L, m, w, uvd, dev((B, J, P, P), seed + 5), dev((B, J, P, P), seed + 7), dev((B, J, 3), seed + 6), method)
        return [p, uvd, gz, gD] + ([gw] if gw is not None else [])
    return run


def _pool(B, H, C, seed):
    def run():
        from pixelwiseregression_amd import kernels as K
        x = dev((B, H, H, C), seed, dtype=BF)
        h = K.maxpool_fwd(x)
        up = K.upsample_add(h, x)
        dh = K.upsample_bwd(up, H // 2, H // 2)
        dx = K.maxpool_bwd(x, dh, addend=up)
        return [h, up, dh, dx, K.colsum_nhwc(x)]
    return run


def _cat(B, J, P, seed):
    def run():
        from pixelwiseregression_amd import kernels as K
        pm, dm, lb = dev((B, J, P, P), seed).abs(), dev((B, J, P, P), seed + 1), dev((B, 1, P, P), seed + 2)
        xc = K.cat_to_nhwc(pm, dm, lb, BF)
        gp, gd = K.nhwc_to_cat_grad(xc, J)
        return [xc, gp, gd, K.nchw_to_nhwc_pad(pm, 16, BF), K.planesum_nchw(pm)]
    return run


def _stem(B, S, seed):
    def run():
        from pixelwiseregression_amd import kernels as K
        img, w, b = dev((B, 1, S, S), seed), dev((32, 1, 3, 3), seed + 1, 0.4), dev((32,), seed + 2, 0.1)
        y = K.stem_conv_fwd(img, w, b, BF)
        return [y, K.stem_conv_wgrad(img, dev((B, S, S, 32), seed + 3, dtype=BF), 3)]
    return run


def _train_loss():
    """220 AdamW steps of the BASELINE C2 bench configuration (tools/lease_check.py's first check): the final loss, bit for bit."""
    def run():
        from pixelwiseregression_amd import PixelwiseRegression
        from pixelwiseregression_amd.synthetic import make_batch
        from pixelwiseregression_amd.train import TrainStep
        torch.manual_seed(0)
        m = PixelwiseRegression(14, stage=2, label_size=64, features=128, level=4, norm_method="instance").to(DEV).set_precision("bf16").train()
        tr = TrainStep(m, opt="adam", lr=1e-4, beta1=0.9, beta2=0.999, weight_decay=0.0, alpha=1.0, lambda_h=1.0, lambda_d=0.01)
        b = make_batch(32, 14, S=128, seed=1234, device=DEV, dense_targets=True)
        for _ in range(220):
            loss = tr(b["img"], b["label_img"], b["mask"], b["uvd"], b["heatmaps"], b["depthmaps"])
        return [loss.detach().double().reshape(1)]
    return run


# name -> thunk returning the list of output tensors; the name starts with the kernel family (what a mismatch should be read as)
CASES = {
    # conv3x3_patch_kernel<bf16,128,...,MF=16> (the dominant kernel) and its pair / statistics / data-gradient forms
    "conv3x3_patch/128to128_fwd_norm_prologue": _conv(3, 64, 64, 128, 128, 3, seed=100),
    "conv3x3_patch/128to128_fwd_plain": _conv(2, 32, 96, 128, 128, 3, prologue=False, seed=110),
    "conv3x3_patch/128to128_fwd_residual": _conv(2, 32, 32, 128, 128, 3, residual=True, seed=120),
    "conv3x3_patch/128to128_fwd_stats": _conv_stats(3, 64, 64, 128, 128, 3, seed=130),
    "conv3x3_patch/128to128_dgrad_nbsums": _dgrad_stats(3, 64, 64, 128, 128, 3, seed=140),
    "conv3x3_patch_pair/128to128_fwd_stats": _pair_fwd(3, 64, 128, seed=150),
    "conv3x3_patch_pair/128to128_dgrad_nbsums": _pair_dgrad(3, 64, 128, seed=170),
    "conv3x3_patch_pair/32to128_dgrad_nbsums": _pair_dgrad(2, 64, 32, seed=190),
    # the other tiles of the patch kernel
    "conv3x3_patch/64to128_at128": _conv(1, 128, 128, 64, 128, 3, seed=200),
    "conv3x3_patch/32to64_at128_stats": _conv_stats(1, 128, 128, 32, 64, 3, seed=210),
    "conv3x3_patch/64to64_at32": _conv(3, 32, 32, 64, 64, 3, seed=220),
    "conv3x3_patch/128to64_dgrad_nbsums": _dgrad_stats(1, 128, 128, 128, 64, 3, seed=225),
    "conv3x3_patch/128toJ_nchw": _conv_nchw(2, 64, 128, 14, seed=230),
    "conv3x3_patch/stride2_fwd": _conv_stats(2, 128, 128, 128, 128, 3, seed=240, stride=2),
    "conv3x3_patch/stride2_dgrad_nbsums": _dgrad_stats(2, 64, 64, 128, 128, 3, seed=250, mode=1),
    "conv3x3_patch/1x1_128to64": _conv(2, 64, 64, 128, 64, 1, seed=260),
    "conv3x3_patch/1x1_64to128_residual_stats": _conv(2, 32, 32, 64, 128, 1, residual=True, seed=270),
    "conv3x3_patch/small_16": _conv(3, 16, 16, 64, 64, 3, seed=280),
    "conv3x3_patch/small_8": _conv(5, 8, 8, 64, 64, 3, seed=290),
    "conv3x3_patch/small_4": _conv(9, 4, 4, 64, 64, 3, seed=300),
    "conv3x3_patch/small_2": _conv(40, 2, 2, 64, 64, 3, seed=310),
    # universal implicit GEMM (5x5, ragged maps, fp32 parity mode)
    "conv_fwd_kernel/5x5_bf16": _conv(2, 32, 32, 128, 128, 5, seed=320),
    "conv_fwd_kernel/ragged_bf16": _conv(3, 5, 7, 64, 64, 3, seed=330),
    "conv_fwd_kernel/fp32_128to128": _conv(1, 32, 32, 128, 128, 3, seed=340, dtype=torch.float32),
    # weight gradients
    "conv_wgrad3w/128to128_norm_24splits": _wgrad(4, 64, 64, 128, 128, 3, 1, 24, True, seed=400),
    "conv_wgrad3w/128to128_plain_80splits": _wgrad(4, 64, 64, 128, 128, 3, 1, 80, False, seed=410),
    "conv_wgrad3w/pair_norm_24splits": _wgrad_pair(3, seed=420),
    "conv_wgrad3_KPX64/128toJ_norm": _wgrad(4, 64, 64, 128, 16, 3, 1, 64, True, seed=440, cout_real=14),
    "conv_wgrad3_KPX64/64to64_norm": _wgrad(3, 64, 64, 64, 64, 3, 1, 9, True, seed=450),
    "conv_wgrad3_KPX64/empty_splits": _wgrad(18, 64, 64, 128, 16, 3, 1, 256, True, seed=455, cout_real=14),
    "conv_wgrad3_KPX64/64to128_at128": _wgrad(1, 128, 128, 64, 128, 3, 1, 40, True, seed=460),
    "conv_wgrad3_STR2/128to128": _wgrad(2, 128, 128, 128, 128, 3, 2, 16, True, seed=470),
    "conv_wgrad3d/64to64_at32_plain": _wgrad(4, 32, 32, 64, 64, 3, 1, 40, False, seed=480),
    "conv_wgrad3d/64to64_at32_norm_in_lds": _wgrad(4, 32, 32, 64, 64, 3, 1, 40, True, seed=490),
    "conv_wgrad_tr/1x1_128to64": _wgrad(4, 64, 64, 128, 64, 1, 1, 64, True, seed=500),
    "conv_wgrad_tr/5x5": _wgrad(2, 16, 16, 32, 32, 5, 1, 4, False, seed=510),
    "conv_wgrad_tr_group/small_maps": _wgrad_group(6, seed=520),
    "conv_wgrad/fp32": _wgrad(2, 16, 16, 32, 32, 3, 1, 4, True, seed=530, dtype=torch.float32),
    # one-launch ResBlocks of the small maps
    "resblock_small/16": _resblock(2, 16, seed=600),
    "resblock_small/8": _resblock(5, 8, seed=620),
    "resblock_small/4": _resblock(9, 4, seed=640),
    "resblock_small/2": _resblock(3, 2, seed=660),
    # norms
    "norm/stats_bwd_64x64x128": _norm(3, 64, 64, 128, seed=700),
    "norm/stats_bwd_small": _norm(32, 4, 4, 64, seed=710),
    "norm/stats_bwd_fp32": _norm(2, 16, 16, 32, seed=720, dtype=torch.float32),
    "norm_bwd_from_partial/single_and_pair": _norm_from_partial(3, 64, 128, seed=730),
    "norm_finalize_chunks/128": _finalize(3, 64, 128, seed=780),
    # decoder, data movement, stem
    "decode/softmax_P64": _decoder(3, 14, 64, seed=800),
    "decode/sum_P16": _decoder(2, 4, 16, seed=810, method="sum"),
    "decode/softmax_P128": _decoder(2, 5, 128, seed=820),
    "pool/maxpool_upsample_colsum": _pool(2, 64, 128, seed=900),
    "pool/cat_and_transposes": _cat(2, 14, 64, seed=910),
    "conv_direct/stem_fwd_wgrad": _stem(3, 128, seed=920),
    # the whole train step
    "train_step/220_adamw_steps_final_loss": _train_loss(),
}


def compute(names=None):
    out = {}
    for name, fn in CASES.items():
        if names and name not in names:
            continue
        out[name] = digest(fn())
        torch.cuda.synchronize()
    return out
