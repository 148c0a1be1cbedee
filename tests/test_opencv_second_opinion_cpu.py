"""CPU: independent second opinions for the three OpenCV restatements of oracle/preprocess_ref.py (cv2 itself cannot be installed
in this image, so tests/golden/preprocess.npz pins the reference's logic AROUND those calls and the calls themselves to these
restatements).  Each restatement is compared here with an implementation that shares no code with it:

  resize_linear (cv2.resize, INTER_LINEAR; /root/reference/datasets.py:270, 297)
      vs torch.nn.functional.interpolate(mode='bilinear', align_corners=False, antialias=False) -- the same half-pixel-centre
      geometry -- at every crop-box size the fixture frames produce and at the 2:1 label resize;
  warp_affine (cv2.warpAffine, INTER_LINEAR, BORDER_CONSTANT; utils.py:73)
      vs torch.nn.functional.grid_sample (exact bilinear sampling at the exact source coordinates): equal to float rounding when the
      source coordinates are multiples of OpenCV's 1/32-pixel fixed-point quantum, and within gradient x half a quantum otherwise;
  rotation_matrix (cv2.getRotationMatrix2D; utils.py:72)
      vs the composition T(c) . R(-angle) . S(scale) . T(-c) built from plain numpy matrices (the closed form in OpenCV's documentation).
Residuals are printed with `pytest -s` and quoted in DESIGN.md section 8."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import preprocess_ref as R


def _smooth(n, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:n, 0:n].astype(np.float64)
    img = np.zeros((n, n))
    for _ in range(6):
        kx, ky, ph = rng.uniform(0.02, 0.25, 2).tolist() + [rng.uniform(0, 6.28)]
        img += rng.uniform(0.3, 1.0) * np.sin(kx * xx + ky * yy + ph)
    return img.astype(np.float32)


def test_resize_linear_vs_torch_interpolate(golden_dir):
    g = np.load(os.path.join(golden_dir, "preprocess.npz"))
    boxes = sorted({int(g[k]) for k in g.files if k.endswith("_box_size")})
    assert len(boxes) >= 5
    worst = 0.0
    rng = np.random.default_rng(0)
    for n, d in [(b, 128) for b in boxes] + [(128, 64), (37, 128), (301, 128)]:
        src = (rng.standard_normal((n, n)) * 100).astype(np.float32)       # rough data: every tap matters
        src[rng.random((n, n)) < 0.3] = 0                                     # background zeros like a depth crop
        ours = R.resize_linear(src, (d, d))
        ref = F.interpolate(torch.from_numpy(src)[None, None].double(), size=(d, d), mode="bilinear", align_corners=False, antialias=False)[0, 0].numpy()
        err = np.abs(ours - ref).max() / np.abs(src).max()
        worst = max(worst, err)
        assert err < 4e-7, (n, d, err)                                        # float32 taps (OpenCV) vs float64: a few ulp of the data range
    print("resize_linear vs F.interpolate: worst relative residual %.2e over box sizes %s" % (worst, boxes))


@pytest.mark.parametrize("angle,scale", [(0.0, 1.0), (17.3, 1.0), (-29.0, 0.83), (30.0, 1.19), (5.5, 1.0)])
def test_warp_affine_vs_grid_sample(angle, scale):
    S = 128
    img = _smooth(S, 3)
    M = R.rotation_matrix((S // 2, S // 2), angle, scale)
    ours = R.warp_affine(img, M, (S, S))
    Mi = R.invert_affine(M)
    ys, xs = np.mgrid[0:S, 0:S].astype(np.float64)
    sx = Mi[0, 0] * xs + Mi[0, 1] * ys + Mi[0, 2]
    sy = Mi[1, 0] * xs + Mi[1, 1] * ys + Mi[1, 2]
    # grid_sample with align_corners=True: normalised -1..1 <-> pixel centres 0..S-1; zeros outside (BORDER_CONSTANT 0)
    grid = torch.from_numpy(np.stack([sx / (S - 1) * 2 - 1, sy / (S - 1) * 2 - 1], axis=-1))[None]
    ref = F.grid_sample(torch.from_numpy(img)[None, None].double(), grid, mode="bilinear", padding_mode="zeros", align_corners=True)[0, 0].numpy()
    gy, gx = np.gradient(img.astype(np.float64))
    lip = np.abs(gx).max() + np.abs(gy).max()
    # OpenCV rounds the source coordinate to 1/32 pixel (INTER_BITS = 5): at most half a quantum off per axis (+ the 1/1024 of AB_BITS)
    inner = (sx > 1) & (sx < S - 2) & (sy > 1) & (sy < S - 2)
    err = np.abs(ours - ref)[inner].max()
    bound = lip * (0.5 / 32 + 1.0 / 1024) * 1.05 + 1e-6
    print("warp_affine vs grid_sample, angle %.1f scale %.2f: max residual %.3e (bound from the 1/32-pixel quantum %.3e)" % (angle, scale, err, bound))
    assert err <= bound, (err, bound)
    if angle == 0.0 and scale == 1.0:
        assert np.array_equal(ours, img)


def test_warp_affine_is_exact_on_the_fixed_point_lattice():
    """A pure translation by a multiple of 1/32 pixel has every source coordinate ON OpenCV's fixed-point lattice: then the restatement is
    exact bilinear sampling and must agree with grid_sample to float32 rounding (no quantum slack)."""
    S = 96
    img = _smooth(S, 5)
    tx, ty = 3 + 7 / 32, -2 - 19 / 32
    M = np.array([[1.0, 0.0, tx], [0.0, 1.0, ty]])
    ours = R.warp_affine(img, M, (S, S))
    ys, xs = np.mgrid[0:S, 0:S].astype(np.float64)
    grid = torch.from_numpy(np.stack([(xs - tx) / (S - 1) * 2 - 1, (ys - ty) / (S - 1) * 2 - 1], axis=-1))[None]
    ref = F.grid_sample(torch.from_numpy(img)[None, None].double(), grid, mode="bilinear", padding_mode="zeros", align_corners=True)[0, 0].numpy()
    err = np.abs(ours - ref).max()
    print("warp_affine on the 1/32 lattice vs grid_sample: max residual %.2e" % err)
    assert err < 1e-6


@pytest.mark.parametrize("angle,scale,c", [(0.0, 1.0, (64, 64)), (30.0, 1.2, (64, 64)), (-17.5, 0.8, (10.0, 90.5)), (123.0, 1.0, (0, 0))])
def test_rotation_matrix_vs_composed_transform(angle, scale, c):
    a = np.deg2rad(angle)
    T = lambda tx, ty: np.array([[1, 0, tx], [0, 1, ty], [0, 0, 1.0]])
    # OpenCV's convention: positive angle = counter-clockwise with the origin at the TOP-left, i.e. [[cos, sin], [-sin, cos]] in (x, y-down)
    RS = np.array([[scale * np.cos(a), scale * np.sin(a), 0], [-scale * np.sin(a), scale * np.cos(a), 0], [0, 0, 1.0]])
    comp = (T(c[0], c[1]) @ RS @ T(-c[0], -c[1]))[:2]
    ours = R.rotation_matrix(c, angle, scale)
    assert np.abs(ours - comp).max() < 1e-12
    assert np.abs(ours @ np.array([c[0], c[1], 1.0]) - np.array(c, dtype=np.float64)).max() < 1e-12     # the centre is a fixed point
    assert abs(np.linalg.det(ours[:, :2]) - scale * scale) < 1e-12
