"""Input pipeline on the device -- the batched counterpart of
``HandDataset.process_single_data`` (/root/reference/datasets.py:182-403, process_mode 'uvd'):

    batch = preprocess_batch(depth, joint_uvd, com, cube_size, INTRINSICS["NYU"], augmentation=draw_augmentation(B))
    results = model(batch["img"], batch["label_img"], batch["mask"])

``depth`` is a batch of RAW depth frames that already sits in HBM ([B,H,W] fp32, millimetres, 0 = no measurement); the crop around
the hand's centre of mass, the depth cut, the resize to the network's crop size, the rotation / scale / shift augmentation, the
label image, the mask, the normalisation and the dense training targets are computed there by HIP kernels (csrc/preprocess.hip,
csrc/targets.hip) instead of per sample with numpy + OpenCV in DataLoader workers.  The few per-sample scalars (crop window,
affine matrix) and the [B,J,3] joint transform are float64 host arithmetic exactly as in the reference.

Returned dict = the reference loader's tuple (datasets.py:403): img [B,1,S,S], label_img [B,1,P,P], mask [B,1,P,P], box_size [B],
cube_size [B], com [B,3], uvd [B,J,3], heatmaps / depthmaps [B,J,P,P] (when ``dense_targets``).

Behaviours of the reference that are kept on purpose:
  * the "shift" augmentation moves the COM by (shift_x, shift_y) PIXELS -- HandDataset.uvd2xyz / xyz2uvd return a 1-D vector
    unchanged (datasets.py:85-111), so the intended camera-space shift of datasets.py:235-241 never happens;
  * the rotation angle that is applied is the one utils.random_rotated draws itself (utils.py:70), not the one of datasets.py:225;
  * a joint for which utils.generate_heatmap raises makes the reference fall back to the un-augmented sample (datasets.py:300);
    ``preprocess_batch`` does the same per sample (``fallback`` in the result says which).  generate_heatmap raises only when
    floor(u) + 1 >= P or floor(u) < -P (same for v): numpy WRAPS the negative indices in between, so a joint just left of / above
    the label map keeps the augmented sample, with its 2x2 footprint split between the opposite borders (utils.py:54-57);
  * a sample is dropped by the reference (ValueError -> check_text, datasets.py:159-167, 358-365, 385-390) when generate_heatmap
    raises on the un-augmented path too, when anything is NaN, or when the mask has fewer than 10 pixels: ``rejected`` says which
    samples of the batch those are (their tensors are still filled in; the caller drops them).
The flip augmentation is not offered: in the reference it raises (joint_uvd_centered is used before assignment, datasets.py:268)
and therefore always ends in the un-augmented fallback.
"""
import math
import random as _random

import numpy as np
import torch

from . import _lib
from .targets import make_targets


def draw_augmentation(B, rng=_random, using_rotation=True, using_scale=True, using_shift=True):
    """The reference's random draws, in its order, per sample (datasets.py:224-238, utils.py:70)."""
    out = {"angle": [], "scale": [], "shift_x": [], "shift_y": []}
    for _ in range(B):
        if using_rotation:
            rng.random()                                            # datasets.py:225 (drawn, never used)
        out["scale"].append(0.8 + rng.random() * 0.4 if using_scale else 1.0)
        if using_shift:
            out["shift_x"].append(-5 + rng.random() * 10)
            out["shift_y"].append(-5 + rng.random() * 10)
        else:
            out["shift_x"].append(0.0); out["shift_y"].append(0.0)
        out["angle"].append(rng.random() * 60 - 30)                 # utils.py:70
    return {k: np.asarray(v, dtype=np.float64) for k, v in out.items()}


def _rotation_inverse(S, angle, scale):
    """cv2.getRotationMatrix2D((S//2, S//2), angle, scale), inverted the way cv2.warpAffine inverts it (float64)."""
    a = angle * math.pi / 180.0
    al, be = scale * math.cos(a), scale * math.sin(a)
    cx = cy = float(S // 2)
    M = [[al, be, (1 - al) * cx - be * cy], [-be, al, be * cx + (1 - al) * cy]]
    D = M[0][0] * M[1][1] - M[0][1] * M[1][0]
    D = 1.0 / D if D != 0 else 0.0
    i00, i01, i10, i11 = M[1][1] * D, -M[0][1] * D, -M[1][0] * D, M[0][0] * D
    return [i00, i01, -i00 * M[0][2] - i01 * M[1][2], i10, i11, -i10 * M[0][2] - i11 * M[1][2]]


def _run(depth, com, cube, fx, fy, S, P, aug):
    """One pass of the three kernels.  com [B,3] float64 numpy (already shifted if augmenting).  Returns device tensors + host box/com."""
    l = _lib.lib()
    dev = depth.device
    B, H, W = depth.shape
    geo = np.zeros((B, 3), np.int32)
    box = np.zeros(B, np.int64)
    com_i = com.copy()
    for b in range(B):
        du, dv = cube[b] / com[b, 2] * fx, cube[b] / com[b, 2] * fy
        bs = max(int(du + dv), 2)                                   # datasets.py:243-247
        sh = bs // 2
        geo[b] = (int(com[b, 1]) - sh, int(com[b, 0]) - sh, 2 * sh)  # utils.center_crop, centre = (com_v, com_u)
        box[b] = 2 * sh                                             # datasets.py:258: box_size = crop_img.shape[0]
        com_i[b, 0], com_i[b, 1] = int(com[b, 0]), int(com[b, 1])   # datasets.py:255-256
    stream = _lib.stream_ptr(dev)
    geo_d = torch.from_numpy(geo).to(dev)
    comz_d = torch.from_numpy(np.ascontiguousarray(com[:, 2])).to(dev)
    cube_d = torch.from_numpy(np.asarray(cube, dtype=np.float64)).to(dev)
    img = torch.empty(B, S, S, device=dev, dtype=torch.float32)
    _lib.check(l.pwr_crop_resize(depth.data_ptr(), geo_d.data_ptr(), comz_d.data_ptr(), cube_d.data_ptr(), img.data_ptr(), B, H, W, S, stream),
               "pwr_crop_resize")
    if aug is not None:
        minv = torch.tensor([_rotation_inverse(S, float(aug["angle"][b]), float(aug["scale"][b])) for b in range(B)], dtype=torch.float64).to(dev)
        sc = torch.from_numpy(np.asarray(aug["scale"], dtype=np.float64)).float().to(dev)
        warped = torch.empty_like(img)
        _lib.check(l.pwr_warp_affine(img.data_ptr(), minv.data_ptr(), sc.data_ptr(), warped.data_ptr(), B, S, stream), "pwr_warp_affine")
        img = warped
    cube_f = cube_d.float()
    img_n = torch.empty(B, 1, S, S, device=dev, dtype=torch.float32)
    label_n = torch.empty(B, 1, P, P, device=dev, dtype=torch.float32)
    mask = torch.empty(B, 1, P, P, device=dev, dtype=torch.float32)
    _lib.check(l.pwr_label_mask_normalize(img.data_ptr(), cube_f.data_ptr(), img_n.data_ptr(), label_n.data_ptr(), mask.data_ptr(), B, S, P, stream),
               "pwr_label_mask_normalize")
    return img_n, label_n, mask, box, com_i


def _joints(joint_uvd, com_i, box, cube, S, aug):
    """datasets.py:272-283 / 355-357, 378-381 in float64 on the host ([B,J,3] is a few hundred numbers)."""
    cen = np.asarray(joint_uvd, dtype=np.float64) - com_i[:, None, :]
    cen[:, :, :2] = cen[:, :, :2] / (box[:, None, None] - 1) * (S - 1)
    if aug is not None:
        for b in range(cen.shape[0]):
            a = aug["angle"][b] / 180.0 * np.pi
            Rot = np.array([[np.cos(a), np.sin(a)], [-np.sin(a), np.cos(a)]])
            cen[b, :, :2] = cen[b, :, :2] @ Rot.T
            cen[b, :, :2] = cen[b, :, :2] * aug["scale"][b]
            cen[b, :, 2] *= aug["scale"][b]
    uvd = cen.copy()
    uvd[:, :, :2] /= (S - 1)
    uvd[:, :, 2] /= np.asarray(cube, dtype=np.float64)[:, None]
    return uvd


def _footprint_ok(uvd, P):
    """Per sample: utils.generate_heatmap (utils.py:37-61) does not raise for any joint.  It indexes the map with numpy's rule, so
    negative indices down to -P WRAP to the opposite border instead of failing; it raises only for floor(u) + 1 >= P,
    floor(u) < -P (same for v) or a NaN position."""
    uv = uvd[:, :, :2] * (P - 1) + P // 2
    fin = np.isfinite(uv)
    lo = np.floor(np.where(fin, uv, 0.0))
    return (fin & (lo >= -P) & (lo + 1 < P)).all(axis=(1, 2))


def preprocess_batch(depth, joint_uvd, com, cube_size, intrinsics, image_size=128, label_size=64, augmentation=None, kernel_size=7,
                     sigmoid=1.5, dense_targets=True):
    if not depth.is_cuda:
        raise _lib.PwrError("preprocess_batch needs the raw depth frames on the GPU (there is no CPU path)")
    fx, fy = float(intrinsics[0]), float(intrinsics[1])
    depth = depth.contiguous().float()
    B = depth.shape[0]
    S, P = int(image_size), int(label_size)
    com = np.array(com.cpu() if isinstance(com, torch.Tensor) else com, dtype=np.float64).reshape(B, 3)
    joint_uvd = np.array(joint_uvd.cpu() if isinstance(joint_uvd, torch.Tensor) else joint_uvd, dtype=np.float64)
    cube = np.full(B, float(cube_size)) if np.isscalar(cube_size) else np.asarray(cube_size, dtype=np.float64).reshape(B)
    dev = depth.device
    fallback = np.zeros(B, dtype=bool)
    if augmentation is not None:
        com_a = com.copy()
        com_a[:, 0] += augmentation["shift_x"]                      # (pixels: see the module docstring)
        com_a[:, 1] += augmentation["shift_y"]
        img, label, mask, box, com_i = _run(depth, com_a, cube, fx, fy, S, P, augmentation)
        uvd = _joints(joint_uvd, com_i, box, cube, S, augmentation)
        fallback = ~_footprint_ok(uvd, P)
    if augmentation is None or fallback.any():
        img0, label0, mask0, box0, com0 = _run(depth, com, cube, fx, fy, S, P, None)
        uvd0 = _joints(joint_uvd, com0, box0, cube, S, None)
        if augmentation is None:
            img, label, mask, box, com_i, uvd = img0, label0, mask0, box0, com0, uvd0
        else:                                                       # datasets.py:300: the reference's per-sample fallback
            fb = torch.from_numpy(fallback).to(dev)
            img = torch.where(fb[:, None, None, None], img0, img)
            label = torch.where(fb[:, None, None, None], label0, label)
            mask = torch.where(fb[:, None, None, None], mask0, mask)
            box[fallback], com_i[fallback], uvd[fallback] = box0[fallback], com0[fallback], uvd0[fallback]
    # datasets.py:358-365 (un-augmented heat map fails) and :385-390 (NaN anywhere, or fewer than 10 mask pixels): sample dropped
    # `rejected` stays ON THE DEVICE (no host synchronisation here: the input pipeline must not serialise with the train step that runs
    # beside it; the caller reads the flags when it needs them): the host-side footprint / NaN-joint flags go up, the image-side flags
    # (NaN anywhere, mask.sum() < 10) are computed where the images are
    rejected_host = ~_footprint_ok(uvd, P) | np.isnan(uvd).any(axis=(1, 2))
    bad = torch.isnan(img).flatten(1).any(1) | torch.isnan(label).flatten(1).any(1) | (mask.flatten(1).sum(1) < 10)
    rejected = torch.from_numpy(rejected_host).to(dev, non_blocking=True) | bad
    out = {"img": img, "label_img": label, "mask": mask, "box_size": torch.from_numpy(box.astype(np.float32)),
           "cube_size": torch.from_numpy(cube.astype(np.float32)), "com": torch.from_numpy(com_i.astype(np.float32)),
           "uvd": torch.from_numpy(uvd.astype(np.float32)).to(dev), "fallback": torch.from_numpy(fallback), "rejected": rejected}
    if dense_targets:
        out["heatmaps"], out["depthmaps"] = make_targets(out["uvd"], label, mask, kernel_size, sigmoid)
    return out
