"""GPU parity: the HIP decoder (through the C ABI) against reference-generated golden vectors and the
numpy oracle.  Tolerances: p 2e-6 abs (values <= 1), uvd 1e-5, gradients 1e-5 relative to max |g|
(BASELINE.json asks for 1e-4 end to end)."""
import os

import numpy as np
import pytest
import torch

from oracle import decoder_ref

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).float().to(_dev())


@pytest.mark.parametrize("method", ["softmax", "sum"])
@pytest.mark.parametrize("P", [16, 64])
@pytest.mark.parametrize("mask", ["bin", "soft"])
def test_decoder_golden(golden_dir, method, P, mask):
    from pixelwiseregression_amd import ops
    g = np.load(os.path.join(golden_dir, "decoder_%s_P%d.npz" % (method, P)))
    key = "f32_%s_" % mask
    z, D, L, m, w = _t(g["z"]), _t(g["D"]), _t(g["L"]), _t(g["m_" + mask]), _t(g["w"])
    wq = w if method == "softmax" else None
    p, uvd = ops.decode_forward(z, D, L, m, wq, method)
    torch.cuda.synchronize()
    # compare against the fp64 reference vectors too: the kernel should be at least as close as ATen fp32
    np.testing.assert_allclose(p.cpu().numpy(), g[key + "p"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(p.cpu().numpy(), g["f64_%s_p" % mask], rtol=0, atol=2e-6)
    np.testing.assert_allclose(uvd.cpu().numpy(), g["f64_%s_uvd" % mask], rtol=0, atol=1e-5)
    gz, gD, gw = ops.decode_backward(p, z, D, L, m, wq, uvd, _t(g["gH"]), _t(g["gD"]), _t(g["gU"]), method)
    torch.cuda.synchronize()
    for got, name in ((gz, "gz"), (gD, "gD")):
        ref = g["f64_%s_%s" % (mask, name)]
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=0, atol=1e-5 * max(1.0, np.abs(ref).max()))
    if method == "softmax":
        ref = g["f64_%s_gw" % mask]
        np.testing.assert_allclose(gw.cpu().numpy(), ref, rtol=0, atol=1e-4 * max(1.0, np.abs(ref).max()))


@pytest.mark.parametrize("B,J,P", [(1, 1, 2), (2, 5, 6), (3, 14, 32), (2, 21, 64), (1, 3, 128), (1, 2, 40)])
@pytest.mark.parametrize("method", ["softmax", "sum"])
def test_decoder_vs_oracle_shapes(B, J, P, method):
    """Ragged / odd sizes (generic kernel), the register-resident sizes (32, 64, 128) and edge cases:
    an all-masked sample (denominator is only the 1e-14 guard) and a one-hot-like logit map."""
    from pixelwiseregression_amd import ops
    rng = np.random.default_rng(B * 1000 + J * 10 + P)
    z = rng.standard_normal((B, J, P, P)).astype(np.float32) * 3
    z[0, 0, P // 2, P // 3] = 40.0                      # near one-hot softmax
    D = rng.standard_normal((B, J, P, P)).astype(np.float32)
    m = (rng.random((B, 1, P, P)) < 0.5).astype(np.float32)
    m[-1] = 0.0                                          # fully masked sample
    L = (rng.standard_normal((B, 1, P, P)).astype(np.float32)) * m
    w = (1 + 0.3 * rng.standard_normal((J, 1))).astype(np.float32)
    gH = rng.standard_normal((B, J, P, P)).astype(np.float32)
    gU = rng.standard_normal((B, J, 3)).astype(np.float32)
    p_ref, uvd_ref = decoder_ref.decode_forward(z, D, L, m, w, method, dtype=np.float64)
    gz_ref, gD_ref, gw_ref = decoder_ref.decode_backward(z, D, L, m, w, gH, None, gU, method, dtype=np.float64)
    wq = _t(w) if method == "softmax" else None
    p, uvd = ops.decode_forward(_t(z), _t(D), _t(L), _t(m), wq, method)
    np.testing.assert_allclose(p.cpu().numpy(), p_ref, rtol=0, atol=2e-6)
    np.testing.assert_allclose(uvd.cpu().numpy(), uvd_ref, rtol=0, atol=2e-5)
    assert np.all(uvd.cpu().numpy()[-1, :, 2] == 0.0)   # all-masked: 0 / 1e-14
    gz, gD, gw = ops.decode_backward(p, _t(z), _t(D), _t(L), _t(m), wq, uvd, _t(gH), None, _t(gU), method)
    for got, ref in ((gz, gz_ref), (gD, gD_ref)):
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=0, atol=2e-5 * max(1.0, np.abs(ref).max()))
    if method == "softmax":
        np.testing.assert_allclose(gw.cpu().numpy(), gw_ref, rtol=0, atol=2e-4 * max(1.0, np.abs(gw_ref).max()))


def test_decoder_autograd_matches_torch_ops():
    """DecodeFn under autograd == the same arithmetic written with differentiable torch ops on the GPU."""
    from pixelwiseregression_amd import ops
    from oracle.model_ref import decode_torch
    from pixelwiseregression_amd.model import com_grid
    torch.manual_seed(0)
    B, J, P = 4, 14, 64
    dev = _dev()
    z = torch.randn(B, J, P, P, device=dev, requires_grad=True)
    D = torch.randn(B, J, P, P, device=dev, requires_grad=True)
    m = (torch.rand(B, 1, P, P, device=dev) < 0.4).float()
    L = torch.randn(B, 1, P, P, device=dev) * m
    w = (1 + 0.2 * torch.randn(J, 1, device=dev)).requires_grad_()
    tgt = torch.randn(B, J, 3, device=dev)
    p, uvd = ops.decode(z, D, L, m, w, "softmax")
    loss = ((uvd - tgt) ** 2).sum(dim=2).mean() + 0.5 * (p ** 2).sum(dim=(2, 3)).mean() + 0.01 * (D ** 2).mean()
    gz, gD, gw = torch.autograd.grad(loss, (z, D, w))
    z2, D2, w2 = (t.detach().double().requires_grad_() for t in (z, D, w))
    p2, uvd2 = decode_torch(z2, D2, L.double(), m.double(), w2, com_grid(P).to(dev).double(), "softmax")
    loss2 = ((uvd2 - tgt.double()) ** 2).sum(dim=2).mean() + 0.5 * (p2 ** 2).sum(dim=(2, 3)).mean() + 0.01 * (D2 ** 2).mean()
    gz2, gD2, gw2 = torch.autograd.grad(loss2, (z2, D2, w2))
    assert (uvd - uvd2).abs().max().item() < 1e-5
    for a, b_ in ((gz, gz2), (gD, gD2), (gw, gw2)):
        assert (a.double() - b_).abs().max().item() <= 1e-5 * max(1.0, b_.abs().max().item())


@pytest.mark.parametrize("B,J", [(49, 42), (98, 21), (57, 37)])
def test_decoder_128x128_many_maps_equal_the_same_maps_in_small_launches(B, J):
    """Round 6: with >= 2048 maps of 128x128 a forward workgroup owns several consecutive maps of one sample and software-pipelines them
    (the next map's operands are requested before this map's reductions and stores), and the workgroups are dealt to the XCDs sample by
    sample.  Every map must come out as the SAME BYTES as in a small launch of its own sample (one map per workgroup, no pipeline) -- J = 42
    takes the three-map form, J = 21 too, J = 37 (prime) the one-map form at the XCD-aware order with a padded grid (B = 57: 8 x 8 x J
    blocks) -- forward and backward; and a sample's maps against the numpy oracle."""
    from pixelwiseregression_amd import ops
    P = 128
    g = torch.Generator(device="cpu").manual_seed(B * 100 + J)
    dev = _dev()
    z = (torch.randn(B, J, P, P, generator=g) * 2).to(dev)
    D = torch.randn(B, J, P, P, generator=g).to(dev)
    m = (torch.rand(B, 1, P, P, generator=g) < 0.4).float().to(dev)
    L = (torch.randn(B, 1, P, P, generator=g)).to(dev) * m
    w = (1 + 0.1 * torch.randn(J, 1, generator=g)).to(dev)
    gH, gD = torch.randn(B, J, P, P, generator=g).to(dev), torch.randn(B, J, P, P, generator=g).to(dev)
    gU = torch.randn(B, J, 3, generator=g).to(dev)
    p, uvd = ops.decode_forward(z, D, L, m, w, "softmax")
    gz, gDo, gw = ops.decode_backward(p, z, D, L, m, w, uvd, gH, gD, gU, "softmax")
    for b in (0, 7, 8, B - 1):
        sl = slice(b, b + 1)
        p1, u1 = ops.decode_forward(z[sl].contiguous(), D[sl].contiguous(), L[sl].contiguous(), m[sl].contiguous(), w, "softmax")
        assert torch.equal(p1, p[sl]) and torch.equal(u1, uvd[sl]), b
        gz1, gD1, _ = ops.decode_backward(p1, z[sl].contiguous(), D[sl].contiguous(), L[sl].contiguous(), m[sl].contiguous(), w, u1,
                                          gH[sl].contiguous(), gD[sl].contiguous(), gU[sl].contiguous(), "softmax")
        assert torch.equal(gz1, gz[sl]) and torch.equal(gD1, gDo[sl]), b
    b = B - 1
    pr, ur = decoder_ref.decode_forward(z[b:b + 1].cpu().numpy(), D[b:b + 1].cpu().numpy(), L[b:b + 1].cpu().numpy(), m[b:b + 1].cpu().numpy(),
                                        w.cpu().numpy(), "softmax", dtype=np.float64)
    np.testing.assert_allclose(p[b:b + 1].cpu().numpy(), pr, rtol=0, atol=2e-6)
    np.testing.assert_allclose(uvd[b:b + 1].cpu().numpy(), ur, rtol=0, atol=1e-5)
    assert float(gw.abs().max()) > 0
