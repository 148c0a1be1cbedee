"""Seeded synthetic depth crops with the value distribution of the reference's data pipeline.

The datasets themselves are not available, so inputs imitate what
/root/reference/datasets.py:378-403 hands to the model:

  img        [B,1,S,S]  depth_norm * hand-mask; background exactly 0; values in (-1, 1)
  label_img  [B,1,P,P]  img down-sampled to P = S/2 (datasets.py:297, cv2.resize) -- 2x2 area mean here
  mask       [B,1,P,P]  (label_img != 0).float()                       (datasets.py:298-299)
  uvd        [B,J,3]    targets ~ U(-0.4, 0.4)
  heatmaps / depthmaps  optional dense targets (only needed when alpha < 1, train.py:197-198)

Everything is generated on the CPU from a torch.Generator so that the same seed gives the
same batch on every machine, then moved to ``device``.
"""
import torch


def make_batch(B, J, S=128, seed=1234, device="cpu", dense_targets=False):
    g = torch.Generator().manual_seed(int(seed))
    P = S // 2
    ys, xs = torch.meshgrid(torch.arange(S, dtype=torch.float32), torch.arange(S, dtype=torch.float32),
                            indexing="ij")
    hand = torch.zeros(B, S, S, dtype=torch.bool)
    n_disc = 6
    # union of discs inside the central 70 % of the crop -> roughly 25-45 % coverage
    cx = (0.15 + 0.7 * torch.rand(B, n_disc, generator=g)) * S
    cy = (0.15 + 0.7 * torch.rand(B, n_disc, generator=g)) * S
    rad = (0.10 + 0.10 * torch.rand(B, n_disc, generator=g)) * S
    for k in range(n_disc):
        d2 = (xs[None] - cx[:, k, None, None]) ** 2 + (ys[None] - cy[:, k, None, None]) ** 2
        hand |= d2 < rad[:, k, None, None] ** 2
    # smooth depth: low-resolution noise up-sampled bilinearly, squashed into (-1, 1)
    coarse = torch.randn(B, 1, 8, 8, generator=g)
    depth = torch.nn.functional.interpolate(coarse, size=(S, S), mode="bilinear", align_corners=False)
    depth = torch.tanh(0.8 * depth)
    depth = torch.where(depth == 0, torch.full_like(depth, 1e-3), depth)
    img = depth * hand[:, None].float()
    label_img = torch.nn.functional.avg_pool2d(img, 2)
    mask = (label_img != 0).float()
    uvd = (torch.rand(B, J, 3, generator=g) - 0.5) * 0.8
    out = {"img": img, "label_img": label_img, "mask": mask, "uvd": uvd}
    if dense_targets:
        # soft blobs centred on the target (u,v): stand-in for utils.py:37-65 heatmap targets
        gx = (torch.arange(P, dtype=torch.float32) - P // 2) / (P - 1)
        du = gx[None, None, None, :] - uvd[:, :, 0, None, None]
        dv = gx[None, None, :, None] - uvd[:, :, 1, None, None]
        heat = torch.exp(-(du ** 2 + dv ** 2) / (2 * (1.5 / P) ** 2))
        heat = heat / heat.sum(dim=(2, 3), keepdim=True).clamp_min(1e-12)
        out["heatmaps"] = heat
        out["depthmaps"] = (uvd[:, :, 2, None, None] - label_img) * mask
    return {k: v.to(device) for k, v in out.items()}
