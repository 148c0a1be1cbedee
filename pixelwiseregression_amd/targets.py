"""Dense heat-map / depth-map training targets generated on the GPU (the reference builds them per sample on the CPU:
/root/reference/datasets.py:285-294, :365-383, utils.py:37-64)."""
import torch

from . import _lib


def make_targets(uvd, label_img, mask, kernel_size=7, sigmoid=1.5):
    """uvd [B,J,3] normalised joints, label_img / mask [B,1,P,P] -> (heatmaps, depthmaps), each [B,J,P,P] fp32: what
    HandDataset.__getitem__ returns as `heatmaps` and `normalized_Dmap` (datasets.py:401-403) for the same joints."""
    if not uvd.is_cuda:
        raise _lib.PwrError("make_targets needs GPU tensors (there is no CPU path)")
    uvd, label_img, mask = (t.contiguous().float() for t in (uvd, label_img, mask))
    B, J, _ = uvd.shape
    P = label_img.shape[-1]
    heat = torch.empty(B, J, P, P, device=uvd.device, dtype=torch.float32)
    dmap = torch.empty_like(heat)
    _lib.check(_lib.lib().pwr_make_targets(uvd.data_ptr(), label_img.data_ptr(), mask.data_ptr(), heat.data_ptr(), dmap.data_ptr(),
                                           B, J, P, int(kernel_size), float(sigmoid), _lib.stream_ptr(uvd.device)), "pwr_make_targets")
    return heat, dmap
