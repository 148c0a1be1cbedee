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


# ------------------------------------------------------------------------------------------------------------------
# A LEARNABLE synthetic task: rendered "hands" whose joint targets are a function of the image.
# make_batch() above draws the targets independently of the image (fine for timing and parity, useless for accuracy:
# nothing but the mean can be learnt).  make_pose_batch() renders a kinematic template -- a palm and five fingers as
# capsules with a cylindrical depth profile -- under a random in-plane rotation, scale, shift, tilt and per-finger
# curl, and returns what /root/reference/datasets.py:378-403 hands to the training loop:
#   img, label_img, mask, box_size, cube_size, com, uvd (normalised as in datasets.py:378-381), so that the metric tail
#   of train.py:254-285 (recover_uvd -> uvd2xyz -> mean joint error in mm) applies unchanged.
# The few random pose parameters come from a CPU torch.Generator (same seed -> same batch on every machine); the
# rendering is elementwise torch math on `device`.
# ------------------------------------------------------------------------------------------------------------------
def hand_template(J):
    """Template joint positions in hand units (palm centre at the origin, fingers pointing 'up' = -y) and the index of each
    joint's parent along its finger (-1: the palm centre)."""
    import math
    n_f = 5
    per = max(1, -(-(J - 1) // n_f))          # joints per finger
    ang = [-62.0, -26.0, 0.0, 24.0, 52.0]
    reach = [0.62, 0.92, 1.0, 0.93, 0.78]
    pos, parent = [(0.0, 0.0)], [-1]
    # ordered ring by ring (all first knuckles, then all second joints, ...), truncated to J
    for k in range(per):
        for f in range(n_f):
            if len(pos) >= J:
                break
            r = reach[f] * (0.42 + 0.58 * (k + 1) / per)
            a = math.radians(ang[f])
            pos.append((r * math.sin(a), -r * math.cos(a)))
            parent.append(0 if k == 0 else 1 + (k - 1) * n_f + f)
    return torch.tensor(pos, dtype=torch.float32), parent


def make_pose_batch(B, J, S=128, seed=0, device="cpu", dataset="NYU", cube_size=150.0):
    from .metric import INTRINSICS
    g = torch.Generator().manual_seed(int(seed))
    P = S // 2
    fx, fy, hu, hv = INTRINSICS[dataset]
    tmpl, parent = hand_template(J)
    n_f = 5
    finger_of = [-1] + [(j - 1) % n_f for j in range(1, J)]
    ring_of = [0] + [(j - 1) // n_f + 1 for j in range(1, J)]
    rnd = lambda *s: torch.rand(*s, generator=g)
    theta = (rnd(B) * 60 - 30) * (3.14159265 / 180)                  # in-plane rotation, like datasets.py:225
    scale = 0.46 + 0.12 * rnd(B)                                     # hand length as a fraction of the crop
    shift = (rnd(B, 2) - 0.5) * 0.12                                 # palm centre off the crop centre
    tilt = (rnd(B, 2) - 0.5) * 0.8                                   # depth gradient across the hand
    curl = 0.55 + 0.45 * rnd(B, n_f)                                 # per-finger length factor (1 = straight)
    bend = (rnd(B, n_f) - 0.5) * (20 * 3.14159265 / 180)             # per-finger in-plane deviation
    fo = torch.tensor([f if f >= 0 else 0 for f in finger_of])
    is_f = torch.tensor([1.0 if f >= 0 else 0.0 for f in finger_of])
    # joint positions in hand units: finger joints scaled by the finger's curl and rotated by its bend about the palm centre
    pj = tmpl[None].repeat(B, 1, 1)
    cf = torch.where(is_f[None] > 0, curl[:, fo], torch.ones(B, J))
    bf = bend[:, fo] * is_f[None]
    x = pj[..., 0] * cf
    y = pj[..., 1] * cf
    xr = x * torch.cos(bf) - y * torch.sin(bf)
    yr = x * torch.sin(bf) + y * torch.cos(bf)
    # curled fingers come towards the camera (smaller depth), more so at the tip
    ring = torch.tensor(ring_of, dtype=torch.float32)
    dz = -(1.0 - cf) * 0.9 * ring[None] / max(1.0, float(max(ring_of)))
    ct, st = torch.cos(theta)[:, None], torch.sin(theta)[:, None]
    xs = (xr * ct - yr * st) * scale[:, None] + shift[:, None, 0]     # crop units, centre = 0, range about +-0.5
    ys = (xr * st + yr * ct) * scale[:, None] + shift[:, None, 1] + 0.17
    d = tilt[:, None, 0] * xs + tilt[:, None, 1] * ys + dz * 0.5 + (rnd(B, 1) - 0.5) * 0.2
    uvd = torch.stack([xs, ys, d], dim=2)                             # normalised like datasets.py:378-381
    # ---- render on the device: capsules parent -> joint, cylindrical depth profile, nearest surface wins
    uvd_d = uvd.to(device)
    px = ((torch.arange(S, dtype=torch.float32, device=device) - S // 2) / (S - 1))
    X = px[None, None, None, :]
    Y = px[None, None, :, None]
    a_idx = torch.tensor([p if p >= 0 else 0 for p in parent], device=device)
    A = uvd_d[:, a_idx]                                               # [B,J,3] segment start (the palm's own segment is a point)
    Bv = uvd_d
    ax, ay, az = A[..., 0, None, None], A[..., 1, None, None], A[..., 2, None, None]
    ex, ey, ez = (Bv[..., 0] - A[..., 0])[..., None, None], (Bv[..., 1] - A[..., 1])[..., None, None], (Bv[..., 2] - A[..., 2])[..., None, None]
    den = (ex * ex + ey * ey).clamp_min(1e-12)
    t = (((X - ax) * ex + (Y - ay) * ey) / den).clamp(0, 1)
    dist2 = (X - ax - t * ex) ** 2 + (Y - ay - t * ey) ** 2           # [B,J,S,S]
    rad = torch.full((J,), 0.055, device=device)
    rad[0] = 0.19                                                     # the palm
    rad = rad[None, :, None, None] * (scale.to(device) / 0.5)[:, None, None, None]
    inside = dist2 < rad * rad
    surf = az + t * ez - 0.6 * torch.sqrt((rad * rad - dist2).clamp_min(0))
    surf = torch.where(inside, surf, torch.full_like(surf, 1e9))
    depth = surf.min(dim=1).values                                    # [B,S,S]
    hand = depth < 1e8
    depth = depth.clamp(-0.95, 0.95)
    depth = torch.where(depth == 0, torch.full_like(depth, 1e-3), depth)
    img = (depth * hand.float())[:, None]
    label_img = torch.nn.functional.avg_pool2d(img, 2)
    mask = (label_img != 0).float()
    com_z = 550.0 + 300.0 * rnd(B)
    com = torch.stack([hu + (rnd(B) - 0.5) * 200, hv + (rnd(B) - 0.5) * 150, com_z], dim=1)
    box = (cube_size / com_z * fx).int() + (cube_size / com_z * fy).int()        # datasets.py:243-246
    return {"img": img, "label_img": label_img, "mask": mask, "uvd": uvd_d, "box_size": box.float(),
            "cube_size": torch.full((B,), float(cube_size)), "com": com}


def make_raw_frames(B, J, H=480, W=640, seed=0, device="cpu", dataset="NYU", cube_size=150.0):
    """RAW depth frames as a depth camera delivers them (what datasets.py:182-199 loads from disk): [B,H,W] fp32 millimetres, 0 = no
    measurement -- a hand-sized blob of smooth depth around the centre of mass, far clutter behind it -- with joints inside the blob
    (uvd in pixels / mm) and the COM: the input of preprocess_batch.  Seeded, generated on the CPU, moved to `device`."""
    from .metric import INTRINSICS
    g = torch.Generator().manual_seed(int(seed))
    fx, fy, hu, hv = INTRINSICS[dataset]
    rnd = lambda *s_: torch.rand(*s_, generator=g)
    cz = 550.0 + 300.0 * rnd(B)
    cu, cv = 200 + 240 * rnd(B), 150 + 180 * rnd(B)
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    rad = cube_size / cz * fx * (0.45 + 0.2 * rnd(B))
    dx, dy = xx[None] - cu[:, None, None], yy[None] - cv[:, None, None]
    blob = dx * dx + dy * dy < (rad * rad)[:, None, None]
    depth = torch.where(blob, cz[:, None, None] + 60 * torch.sin(xx / 17.0)[None] * torch.cos(yy / 23.0)[None] + 0.1 * dx, torch.zeros(()))
    clutter = (dx > 0.5 * rad[:, None, None]) & (dy > 0) & (~blob) & (dx < 1.4 * rad[:, None, None])
    depth = torch.where(clutter, cz[:, None, None] + 400.0, depth)           # cut by the cube (datasets.py:251)
    ang, rr = rnd(B, J) * 6.2831853, rad[:, None] * 0.5 * torch.sqrt(rnd(B, J))
    joints = torch.stack([cu[:, None] + rr * torch.cos(ang), cv[:, None] + rr * torch.sin(ang), cz[:, None] + 50 * (rnd(B, J) - 0.5)], dim=2)
    com = torch.stack([cu + 3 * (rnd(B) - 0.5), cv + 3 * (rnd(B) - 0.5), cz], dim=1)
    return {"depth": depth.to(device), "joint_uvd": joints.double(), "com": com.double(), "cube_size": float(cube_size)}


def joint_error_mm(uvd_pred, batch, dataset="NYU"):
    """Per-sample mean 3D joint error in mm of normalised predictions [B,J,3] against the batch's targets, exactly the
    validation metric of train.py:254-285: recover_uvd -> uvd2xyz -> mean over joints of the Euclidean distance."""
    from .metric import INTRINSICS, recover_uvd, uvd2xyz, mean_joint_error
    intr = INTRINSICS[dataset]
    box, cube, com = batch["box_size"].cpu(), batch["cube_size"].cpu(), batch["com"].cpu()
    true_xyz = uvd2xyz(recover_uvd(batch["uvd"].detach().float().cpu().clone(), box, com, cube).numpy(), *intr)
    pred_xyz = uvd2xyz(recover_uvd(uvd_pred.detach().float().cpu().clone(), box, com, cube).numpy(), *intr)
    return mean_joint_error(pred_xyz, true_xyz)
