"""Oracle (test infrastructure): numpy restatement of the metric tail.

  recover_uvd ...... /root/reference/utils.py:332-337   (normalised uvd -> pixel / mm uvd)
  uvd2xyz .......... /root/reference/datasets.py:100-111 (pin-hole back-projection)
  mean joint error . /root/reference/train.py:276,285    (mean_J ||xyz - xyz_gt||_2, then dataset mean)

Unlike the reference's recover_uvd this does NOT mutate its argument.
Never imported by the product package.
"""
import numpy as np

# (fx, fy, halfu, halfv) from datasets.py:406 (MSRA), :521 (ICVL), :693 (NYU), :861 (HAND17)
INTRINSICS = {
    "MSRA": (241.42, 241.42, 160.0, 120.0),
    "ICVL": (241.42, 241.42, 160.0, 120.0),
    "NYU": (588.037, 587.075, 320.0, 240.0),
    "HAND17": (475.065948, 475.065857, 315.944855, 245.287079),
}


def recover_uvd(uvd, box_size, com, cube_size):
    """uvd [B,J,3] float32; box_size [B]; com [B,3]; cube_size [B] -> float32 [B,J,3]."""
    uvd = np.asarray(uvd, dtype=np.float32).copy()
    box_size = np.asarray(box_size, dtype=np.float32)
    cube_size = np.asarray(cube_size, dtype=np.float32)
    com = np.asarray(com, dtype=np.float32)
    uvd[:, :, :2] = uvd[:, :, :2] * (box_size - 1).reshape(-1, 1, 1)
    uvd[:, :, 2] = uvd[:, :, 2] * cube_size[:, None]
    return uvd + com[:, None, :]


def uvd2xyz(uvd, fx, fy, halfu, halfv):
    x = np.array(uvd, copy=True)
    x[..., 0] = (x[..., 0] - halfu) / fx * x[..., 2]
    x[..., 1] = (x[..., 1] - halfv) / fy * x[..., 2]
    return x


def mean_joint_error(xyz, xyz_gt):
    """Per-sample mean over joints of the L2 distance (train.py:276); caller averages samples."""
    return np.mean(np.sqrt(np.sum((xyz - xyz_gt) ** 2, axis=2)), axis=1)
