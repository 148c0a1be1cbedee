"""Metric tail of the reference: normalised uvd -> camera-space mm error.

Mirrors ``utils.recover_uvd`` (utils.py:332-337), ``HandDataset.uvd2xyz`` (datasets.py:100-111) and the
per-sample mean joint error of train.py:276.  These run on [B,J,3] tensors (a few KB) right after the
``.cpu()`` of the model output (train.py:271, test.py:106), so they stay host-side like in the reference.
"""
import numpy as np
import torch

# (fx, fy, halfu, halfv): datasets.py:406 (MSRA), :521 (ICVL), :693 (NYU), :861 (HAND17)
INTRINSICS = {
    "MSRA": (241.42, 241.42, 160.0, 120.0),
    "ICVL": (241.42, 241.42, 160.0, 120.0),
    "NYU": (588.037, 587.075, 320.0, 240.0),
    "HAND17": (475.065948, 475.065857, 315.944855, 245.287079),
}


def recover_uvd(uvd, box_size, com, threshold):
    """Same signature and in-place behaviour as the reference (it mutates ``uvd`` and returns a new tensor)."""
    uvd[:, :, :2] = uvd[:, :, :2] * (box_size - 1).view(-1, 1, 1)
    uvd[:, :, 2] = uvd[:, :, 2] * threshold.unsqueeze(1)
    return uvd + com.unsqueeze(1)


def uvd2xyz(data, fx, fy, halfu, halfv):
    x = data.copy() if isinstance(data, np.ndarray) else data.clone()
    x[..., 0] = (x[..., 0] - halfu) / fx * x[..., 2]
    x[..., 1] = (x[..., 1] - halfv) / fy * x[..., 2]
    return x


def mean_joint_error(xyz, xyz_gt):
    """[B,J,3] x2 -> [B] mean over joints of the Euclidean distance (train.py:276)."""
    if isinstance(xyz, torch.Tensor):
        return torch.sqrt(((xyz - xyz_gt) ** 2).sum(dim=2)).mean(dim=1)
    return np.mean(np.sqrt(np.sum((xyz - xyz_gt) ** 2, axis=2)), axis=1)
