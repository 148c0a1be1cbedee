"""CPU: the oracle's restatement of the reference's input pipeline (oracle/preprocess_ref.py) against fixtures produced by the
reference's own HandDataset.process_single_data (oracle/gen_golden.py::gen_preprocess; cv2 stubbed, see there), plus the host
side of pixelwiseregression_amd.preprocess (draw order, crop geometry, joint transforms)."""
import os
import random

import numpy as np
import pytest

from oracle import preprocess_ref as R, targets_ref as T


def _g(golden_dir):
    return np.load(os.path.join(golden_dir, "preprocess.npz"))


@pytest.mark.parametrize("i", range(6))
@pytest.mark.parametrize("kind", ["plain", "aug"])
def test_oracle_pipeline_vs_reference(golden_dir, kind, i):
    g = _g(golden_dir)
    fx, fy, hu, hv = g["intrinsics"]
    depth, joints, com = g["raw%d_depth" % i], g["raw%d_joints" % i], g["raw%d_com" % i]
    pre = "%s%d_" % (kind, i)
    if kind == "aug":
        a = R.draws_to_augmentation(g[pre + "draws"])
        o = R.process_single(depth, joints, R.shift_com(com, a["shift_x"], a["shift_y"]), 150, fx, fy, 128, 64, angle=a["angle"], scale=a["scale"])
        assert not np.array_equal(g[pre + "img"], g["plain%d_img" % i]), "the fixture really took the augmented path"
    else:
        assert len(g[pre + "draws"]) == 0
        o = R.process_single(depth, joints, com, 150, fx, fy, 128, 64)
    assert o["box_size"] == int(g[pre + "box_size"])
    np.testing.assert_allclose(o["com"], g[pre + "com"], atol=1e-4)
    np.testing.assert_allclose(np.asarray(o["img"], np.float64)[None], g[pre + "img"], atol=5e-7)
    np.testing.assert_allclose(o["label_img"][None], g[pre + "label_img"], atol=5e-7)
    assert np.array_equal(o["mask"][None], g[pre + "mask"])
    np.testing.assert_allclose(o["uvd"], g[pre + "uvd"], atol=1e-7)
    h, d = T.make_targets(o["uvd"][None].astype(np.float32), o["label_img"][None, None].astype(np.float32), o["mask"][None, None].astype(np.float32))
    np.testing.assert_allclose(h[0], g[pre + "heatmaps"], atol=1e-7)
    np.testing.assert_allclose(d[0], g[pre + "depthmaps"], atol=5e-7)


@pytest.mark.parametrize("i", [6, 7])
@pytest.mark.parametrize("kind", ["plain", "aug"])
def test_oracle_pipeline_vs_reference_wrapped_footprint(golden_dir, kind, i):
    """Frames 6 / 7: joint 0 sits at label pixel (-0.5, -0.4) on the un-augmented path / (-0.6, -0.3) on the augmented one.  The
    reference does not fail there (numpy wraps the negative indices, utils.py:54-57): it keeps the sample -- the AUGMENTED one on the
    augmented path -- with the joint's heat map on the four corners."""
    g = _g(golden_dir)
    fx, fy, hu, hv = g["intrinsics"]
    depth, joints, com = g["raw%d_depth" % i], g["raw%d_joints" % i], g["raw%d_com" % i]
    pre = "%s%d_" % (kind, i)
    aug = R.draws_to_augmentation(g[pre + "draws"]) if kind == "aug" else None
    o, fallback, rejected = R.sample_like_reference(depth, joints, com, 150, fx, fy, 128, 64, aug)
    assert not rejected and not bool(g[pre + "rejected"])
    assert fallback == (kind == "aug" and np.array_equal(g[pre + "img"], g["plain%d_img" % i]))
    np.testing.assert_allclose(np.asarray(o["img"], np.float64)[None], g[pre + "img"], atol=5e-7)
    np.testing.assert_allclose(o["uvd"], g[pre + "uvd"], atol=1e-7)
    h, d = T.make_targets(o["uvd"][None].astype(np.float32), o["label_img"][None, None].astype(np.float32), o["mask"][None, None].astype(np.float32))
    np.testing.assert_allclose(h[0], g[pre + "heatmaps"], atol=3e-7)     # (the position goes through float32 here, float64 in the reference)
    np.testing.assert_allclose(d[0], g[pre + "depthmaps"], atol=5e-7)
    if (kind, i) in (("plain", 6), ("aug", 7)):
        uv = T.label_pixels(o["uvd"], 64)[0]
        assert -1 < uv[0] < 0 and -1 < uv[1] < 0
        assert not fallback and h[0, 0, 0, 0] > 0 and h[0, 0, -1, -1] > 0 and h[0, 0, 32, 32] == 0


@pytest.mark.parametrize("kind", ["plain", "aug"])
def test_oracle_rejects_what_the_reference_rejects(golden_dir, kind):
    """Frame 8 (a hand of nine pixels): the reference raises ValueError (sum(mask) < 10, datasets.py:385-390) on both paths; every
    other frame is accepted."""
    g = _g(golden_dir)
    fx, fy, hu, hv = g["intrinsics"]
    n = int(g["n_frames"])
    assert n == 9
    for i in range(n):
        pre = "%s%d_" % (kind, i)
        aug = R.draws_to_augmentation(g[pre + "draws"]) if kind == "aug" else None
        o, fallback, rejected = R.sample_like_reference(g["raw%d_depth" % i], g["raw%d_joints" % i], g["raw%d_com" % i], 150, fx, fy, 128, 64, aug)
        assert rejected == bool(g[pre + "rejected"]) == (i == 8), (kind, i)


def test_draw_augmentation_replays_the_reference_draw_order(golden_dir):
    from pixelwiseregression_amd.preprocess import draw_augmentation
    g = _g(golden_dir)
    for i in range(6):
        random.seed(1000 + i)                      # the seed gen_golden.py used for sample i
        a = draw_augmentation(1, rng=random)
        ref = R.draws_to_augmentation(g["aug%d_draws" % i])
        for k in ("angle", "scale", "shift_x", "shift_y"):
            assert a[k][0] == ref[k], (i, k)


def test_resize_by_two_is_the_2x2_mean_and_warp_identity():
    rng = np.random.default_rng(0)
    a = rng.random((128, 128)).astype(np.float32)
    b = R.resize_linear(a, (64, 64))
    m = (a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2]) / 4
    assert np.abs(b - m).max() < 1e-6
    w = R.warp_affine(a, R.rotation_matrix((64, 64), 0.0, 1.0), (128, 128))
    assert np.array_equal(w, a)
