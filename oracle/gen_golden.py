#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (build container only).

    python oracle/gen_golden.py            # needs /root/reference; writes tests/golden/

The reference is imported in place (never copied): ``model.py`` needs ``utils.py`` which imports
``cv2`` at top level (utils.py:4) -- an empty stub module satisfies it because the two helpers the
model uses (generate_com_filter, xavier_weights_init) never touch cv2.  ``datasets.py`` (for
``uvd2xyz``) additionally wants ``torchvision`` and ``ray`` stubs.  Only inputs, seeds and the
reference's numeric outputs are stored -- no reference source text.

The fixtures are what pins the oracle (oracle/*.py) and, on the GPU box, the HIP path.
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def import_reference():
    sys.dont_write_bytecode = True
    try:                                # a real OpenCV, where there is one, is used as it is (none in the build image)
        import cv2                      # noqa: F401
    except ImportError:
        sys.modules["cv2"] = types.ModuleType("cv2")
    tv = types.ModuleType("torchvision")
    sys.modules.setdefault("torchvision", tv)
    ray = types.ModuleType("ray")
    ray.remote = lambda x: x
    sys.modules.setdefault("ray", ray)
    sys.path.insert(0, REF)
    import model as ref_model          # noqa: E402
    import utils as ref_utils          # noqa: E402
    try:
        import datasets as ref_datasets  # noqa: E402
    except Exception as e:             # pragma: no cover
        print("datasets import failed:", e)
        ref_datasets = None
    return ref_model, ref_utils, ref_datasets


class _Const(torch.nn.Module):
    """Stands in for the conv head so the reference decoder can be driven with chosen logits."""

    def __init__(self, t):
        super().__init__()
        self.t = t

    def forward(self, _):
        return self.t + 0      # fresh tensor: the 'sum' path applies an in-place ReLU


def gen_decoder(ref_model):
    rng = np.random.default_rng(20260101)
    for method in ("softmax", "sum"):
        for P in (16, 64):
            B, J = 2, 3
            z = (rng.standard_normal((B, J, P, P)) * 2.0)
            Dm = rng.standard_normal((B, J, P, P)) * 0.3
            L = np.tanh(rng.standard_normal((B, 1, P, P)))
            m_bin = (rng.random((B, 1, P, P)) < 0.4).astype(np.float64)
            m_bin[1, 0, : P // 2] = 0.0
            L = L * m_bin
            m_soft = rng.random((B, 1, P, P)) * m_bin          # non-binary mask: m enters squared
            w = 1.0 + 0.5 * rng.standard_normal((J, 1))
            gH = rng.standard_normal((B, J, P, P)) * 0.1
            gD = rng.standard_normal((B, J, P, P)) * 0.1
            gU = rng.standard_normal((B, J, 3))
            rec = dict(z=z, D=Dm, L=L, m_bin=m_bin, m_soft=m_soft, w=w, gH=gH, gD=gD, gU=gU)
            for tname, tdt in (("f32", torch.float32), ("f64", torch.float64)):
                for mname, mk in (("bin", m_bin), ("soft", m_soft)):
                    zt = torch.tensor(z, dtype=tdt, requires_grad=True)
                    Dt = torch.tensor(Dm, dtype=tdt, requires_grad=True)
                    plane = ref_model.PlaneRegression(8, J, P, normalization_method=method)
                    depth = ref_model.DepthRegression(8, J)
                    plane = plane.to(tdt)
                    plane.conv = _Const(zt)
                    depth.conv = _Const(Dt)
                    if method == "softmax":
                        plane.w.data = torch.tensor(w, dtype=tdt)
                    # the grid buffer must stay the fp32-rounded one the reference registers
                    heat, uv = plane(None)
                    dmap, dd = depth(None, heat, torch.tensor(L, dtype=tdt), torch.tensor(mk, dtype=tdt))
                    uvd = torch.cat([uv, dd], dim=2)
                    loss = (heat * torch.tensor(gH, dtype=tdt)).sum() + (dmap * torch.tensor(gD, dtype=tdt)).sum() \
                        + (uvd * torch.tensor(gU, dtype=tdt)).sum()
                    loss.backward()
                    key = "%s_%s_" % (tname, mname)
                    rec[key + "p"] = heat.detach().numpy()
                    rec[key + "uvd"] = uvd.detach().numpy()
                    rec[key + "gz"] = zt.grad.numpy()
                    rec[key + "gD"] = Dt.grad.numpy()
                    if method == "softmax":
                        rec[key + "gw"] = plane.w.grad.numpy()
                    rec["grid"] = plane.filter.float().numpy()
            np.savez_compressed(os.path.join(OUT, "decoder_%s_P%d.npz" % (method, P)), **rec)
            print("decoder", method, P)


def _run_model(ref_model, cfg, sd, batch, alpha, train=True, double=False):
    model = ref_model.PixelwiseRegression(cfg["joints"], **{k: v for k, v in cfg.items() if k != "joints"})
    model.load_state_dict(sd, strict=True)
    model.train(train)
    if double:      # the reference evaluated in float64: the "truth" both fp32 implementations are measured against
        model = model.double()
        batch = {k: v.double() for k, v in batch.items()}
    res = model(batch["img"], batch["label_img"], batch["mask"])
    out = {}
    for s, (p, D, uvd) in enumerate(res):
        out["s%d_p" % s] = p.detach().numpy()
        out["s%d_D" % s] = D.detach().numpy()
        out["s%d_uvd" % s] = uvd.detach().numpy()
    if alpha is None:
        return out, None, model
    loss = 0
    for (p, D, uvd) in res:                                    # train.py:195-205
        hl = 1.0 * torch.mean(torch.sum((p - batch["heatmaps"]) ** 2, dim=(2, 3)))
        dl = 0.01 * torch.mean(torch.sum((D - batch["depthmaps"]) ** 2, dim=(2, 3)))
        ul = torch.mean(torch.sum((uvd - batch["uvd"]) ** 2, dim=2))
        loss = loss + alpha * ul + (1 - alpha) * (hl + dl)
    loss.backward()
    grads = {k: p.grad.detach().numpy() for k, p in model.named_parameters()}
    out["loss"] = np.float64(loss.item())
    return out, grads, model


def gen_tiny(ref_model):
    from weights_util import fill_state_dict
    from pixelwiseregression_amd.synthetic import make_batch
    # kernel_size 5 / 7: the scripts' --filter_size (train.py:47 -> model.py:55-64, :165-182; the hourglass keeps 3x3, model.py:139)
    for norm, method, ks in (("instance", "softmax", 3), ("instance", "sum", 3), ("batch", "softmax", 3),
                             ("instance", "softmax", 5), ("batch", "softmax", 7)):
        if True:
            cfg = dict(joints=4, stage=2, label_size=16, features=32, level=2, kernel_size=ks,
                       norm_method=norm, heatmap_method=method)
            torch.manual_seed(0)
            proto = ref_model.PixelwiseRegression(cfg["joints"], **{k: v for k, v in cfg.items() if k != "joints"})
            sd = fill_state_dict(proto.state_dict(), seed=7)
            batch = make_batch(3, 4, S=32, seed=99, dense_targets=True)
            rec = {"cfg_" + k: np.array(v) for k, v in cfg.items()}
            rec.update({"in_" + k: v.numpy() for k, v in batch.items()})
            rec.update({"sd_" + k: v.numpy() for k, v in sd.items()})
            for alpha in (1.0, 0.5):
                out, grads, model = _run_model(ref_model, cfg, sd, batch, alpha, train=True)
                tag = "a%03d_" % int(alpha * 100)
                rec.update({tag + k: v for k, v in out.items()})
                rec.update({tag + "grad_" + k: v for k, v in grads.items()})
                out64, grads64, _ = _run_model(ref_model, cfg, sd, batch, alpha, train=True, double=True)
                rec.update({tag + "f64_" + k: v for k, v in out64.items()})
                rec.update({tag + "f64_grad_" + k: v.astype(np.float32) for k, v in grads64.items()})
                if norm == "batch" and alpha == 1.0:
                    after = model.state_dict()
                    for k in after:
                        if "running_" in k or "num_batches" in k:
                            rec["after_" + k] = after[k].numpy()
            if norm == "batch":
                with torch.no_grad():
                    out, _, _ = _run_model(ref_model, cfg, sd, batch, None, train=False)
                rec.update({"eval_" + k: v for k, v in out.items()})
            np.savez_compressed(os.path.join(OUT, "tiny_%s_%s%s.npz" % (norm, method, "" if ks == 3 else "_k%d" % ks)), **rec)
            print("tiny", norm, method, ks, "keys", len(rec))


def gen_c1(ref_model):
    """Full-size C1 (ICVL J=16, 128x128, B=1, instance norm, F=128, level 4, stage 2) forward."""
    from weights_util import fill_state_dict
    from pixelwiseregression_amd.synthetic import make_batch
    cfg = dict(joints=16, stage=2, label_size=64, features=128, level=4, kernel_size=3,
               norm_method="instance", heatmap_method="softmax")
    proto = ref_model.PixelwiseRegression(cfg["joints"], **{k: v for k, v in cfg.items() if k != "joints"})
    sd = fill_state_dict(proto.state_dict(), seed=11)
    batch = make_batch(1, 16, S=128, seed=5)
    with torch.no_grad():
        out, _, _ = _run_model(ref_model, cfg, sd, batch, None, train=False)
        out64, _, _ = _run_model(ref_model, cfg, sd, batch, None, train=False, double=True)
    rec = {"cfg_" + k: np.array(v) for k, v in cfg.items()}
    rec["weights_seed"] = np.array(11)
    rec["batch_seed"] = np.array(5)
    rec["in_img"] = batch["img"].numpy()
    rec["in_label_img"] = batch["label_img"].numpy()
    rec["in_mask"] = batch["mask"].numpy()
    for s in range(2):
        rec["s%d_uvd" % s] = out["s%d_uvd" % s]
        p, D = out["s%d_p" % s], out["s%d_D" % s]
        rec["s%d_p_max" % s] = p.max(axis=(2, 3))
        rec["s%d_p_argmax" % s] = p.reshape(1, 16, -1).argmax(axis=2)
        rec["s%d_D_mean" % s] = D.mean(axis=(2, 3))
        rec["s%d_D_std" % s] = D.std(axis=(2, 3))
        rec["s%d_p_sample" % s] = p[:, :, ::8, ::8].copy()
        rec["s%d_D_sample" % s] = D[:, :, ::8, ::8].copy()
        rec["s%d_uvd_f64" % s] = out64["s%d_uvd" % s]
        rec["s%d_p_sample_f64" % s] = out64["s%d_p" % s][:, :, ::8, ::8].copy()
        rec["s%d_D_sample_f64" % s] = out64["s%d_D" % s][:, :, ::8, ::8].copy()
    # parameter census: key names + shapes + element count (the state_dict contract)
    rec["sd_keys"] = np.array(list(sd.keys()))
    rec["sd_numel"] = np.array([v.numel() for v in sd.values()])
    np.savez_compressed(os.path.join(OUT, "c1_full.npz"), **rec)
    print("c1 done; params", sum(p.numel() for p in proto.parameters()))
    # key census for all BASELINE configs and both norms (names/shapes only)
    census = {}
    for name, J, norm in (("nyu_instance", 14, "instance"), ("nyu_batch", 14, "batch"),
                          ("msra_instance", 21, "instance")):
        c = dict(cfg, joints=J, norm_method=norm)
        m = ref_model.PixelwiseRegression(J, **{k: v for k, v in c.items() if k != "joints"})
        s = m.state_dict()
        census[name + "_keys"] = np.array(list(s.keys()))
        census[name + "_shapes"] = np.array([",".join(map(str, v.shape)) for v in s.values()])
    np.savez_compressed(os.path.join(OUT, "state_dict_census.npz"), **census)


def gen_init(ref_model):
    """Initial weights of the reference for a fixed torch seed (construction-order + xavier parity)."""
    rec = {}
    for norm in ("instance", "batch"):
        torch.manual_seed(1234)
        m = ref_model.PixelwiseRegression(4, stage=2, label_size=16, features=32, level=2, kernel_size=3,
                                          norm_method=norm, heatmap_method="softmax")
        sd = m.state_dict()
        rec[norm + "_keys"] = np.array(list(sd.keys()))
        rec[norm + "_sum"] = np.array([float(v.double().sum()) for v in sd.values()])
        rec[norm + "_abs"] = np.array([float(v.double().abs().sum()) for v in sd.values()])
        for k in ("conv.0.weight", "conv.0.bias", "stages.1.conv.weight", "stages.1.conv.bias",
                  "stages.0.hourglass.inner.inner.inner.conv.5.weight", "stages.1.depth_regression.conv.9.weight"):
            rec[norm + "_t_" + k] = sd[k].numpy()
    np.savez_compressed(os.path.join(OUT, "init_seed1234.npz"), **rec)
    print("init done")


def gen_metric(ref_utils, ref_datasets):
    rng = np.random.default_rng(3)
    rec = {}
    from oracle.metric_ref import INTRINSICS
    for name, (fx, fy, hu, hv) in INTRINSICS.items():
        B, J = 4, {"MSRA": 21, "ICVL": 16, "NYU": 14, "HAND17": 21}[name]
        uvd = ((rng.random((B, J, 3)) - 0.5) * 0.9).astype(np.float32)
        uvd_gt = ((rng.random((B, J, 3)) - 0.5) * 0.9).astype(np.float32)
        box = (80 + 60 * rng.random(B)).astype(np.float32)
        cube = np.full(B, 150.0, dtype=np.float32)
        com = np.stack([hu + 40 * rng.standard_normal(B), hv + 40 * rng.standard_normal(B),
                        600 + 200 * rng.random(B)], axis=1).astype(np.float32)
        r = ref_utils.recover_uvd(torch.from_numpy(uvd.copy()), torch.from_numpy(box), torch.from_numpy(com),
                                  torch.from_numpy(cube)).numpy()
        rg = ref_utils.recover_uvd(torch.from_numpy(uvd_gt.copy()), torch.from_numpy(box), torch.from_numpy(com),
                                   torch.from_numpy(cube)).numpy()
        ns = types.SimpleNamespace(fx=fx, fy=fy, halfu=hu, halfv=hv)
        xyz = ref_datasets.HandDataset.uvd2xyz(ns, r)
        xyz_gt = ref_datasets.HandDataset.uvd2xyz(ns, rg)
        err = np.mean(np.sqrt(np.sum((xyz - xyz_gt) ** 2, axis=2)), axis=1)       # train.py:276
        for k, v in dict(uvd=uvd, uvd_gt=uvd_gt, box=box, cube=cube, com=com, rec=r, xyz=xyz, xyz_gt=xyz_gt,
                         err=err).items():
            rec["%s_%s" % (name, k)] = v
    np.savez_compressed(os.path.join(OUT, "metric.npz"), **rec)
    print("metric done")


def gen_targets(ref_utils):
    """Bilinear splat of utils.generate_heatmap (utils.py:37-61) at fixed positions, incl. exact-integer and border cases.
    (The Gaussian blur of utils.generate_kernel needs cv2, which is not installed: not generated here.)"""
    P = 64
    rng = np.random.default_rng(11)
    uv = np.concatenate([rng.random((24, 2)) * (P - 2), np.array([[0.0, 0.0], [10.0, 20.5], [61.999, 3.25], [31.5, 31.5]])])
    heat = np.stack([ref_utils.generate_heatmap(P, float(u), float(v)) for u, v in uv])
    # round 3: positions left of / above the map (numpy wraps the negative indices, utils.py:54-57) and the ones that do raise
    uv_edge = np.array([[-0.5, 10.3], [10.3, -0.5], [-0.25, -0.75], [-1.0, 5.0], [-17.3, 40.2], [-63.5, -63.5], [-64.0, 0.0],
                        [-64.01, 3.0], [3.0, -65.0], [62.99, 62.99], [63.0, 3.0], [3.0, 63.0], [70.0, 3.0], [np.nan, 3.0]])
    splat_edge, raises = [], []
    for u, v in uv_edge:
        try:
            splat_edge.append(ref_utils.generate_heatmap(P, float(u), float(v)))
            raises.append(False)
        except Exception:
            splat_edge.append(np.zeros((P, P)))
            raises.append(True)
    np.savez_compressed(os.path.join(OUT, "targets.npz"), P=np.int64(P), uv=uv, splat=heat, uv_edge=uv_edge, splat_edge=np.stack(splat_edge),
                        raises_edge=np.array(raises))
    print("targets done")


def gen_wellcond(ref_model):
    """WELL-CONDITIONED gradient fixtures: (config, weights seed, rendered-hand batch) triples on which the reference's own
    fp32 gradient is within 6e-5 (per tensor, relative to the tensor's largest entry) of its float64 evaluation -- no ReLU /
    max-pool / arg-max decision sits at a near-tie -- so that a hard per-tensor bound can be asserted on the engine
    (tests/test_parity_holes_gpu.py).  Innermost hourglass maps are 8x8 (a, c) and 4x4 (b)."""
    from weights_util import fill_state_dict
    from pixelwiseregression_amd.synthetic import make_batch, make_pose_batch
    rec = {}
    for tag, (stage, level, alpha) in (("a", (2, 1, 0.5)), ("b", (2, 2, 1.0)), ("c", (1, 1, 0.5))):
        cfg = dict(joints=4, stage=stage, label_size=32, features=32, level=level, kernel_size=3, norm_method="instance",
                   heatmap_method="softmax")
        proto = ref_model.PixelwiseRegression(4, **{k: v for k, v in cfg.items() if k != "joints"})
        sd = fill_state_dict(proto.state_dict(), seed=7)
        batch = make_batch(2, 4, S=64, seed=99, dense_targets=True)
        pb = make_pose_batch(2, 4, 64, seed=5)
        batch.update({k: pb[k] for k in ("img", "label_img", "mask", "uvd")})
        o32, g32, _ = _run_model(ref_model, cfg, sd, batch, alpha)
        o64, g64, _ = _run_model(ref_model, cfg, sd, batch, alpha, double=True)
        gm = max(np.abs(v).max() for v in g64.values())
        worst = max(np.abs(g32[k] - g64[k]).max() / np.abs(g64[k]).max() for k in g32 if np.abs(g64[k]).max() > 1e-6 * gm)
        assert worst < 6e-5, (tag, worst)
        pre = tag + "_"
        rec.update({pre + "cfg_" + k: np.array(v) for k, v in cfg.items()})
        rec[pre + "alpha"] = np.float64(alpha)
        rec[pre + "weights_seed"] = np.array(7)
        rec[pre + "ref32_vs_f64_worst"] = np.float64(worst)
        rec.update({pre + "in_" + k: v.numpy() for k, v in batch.items()})
        rec.update({pre + "f32_" + k: v for k, v in o32.items()})
        rec.update({pre + "f64_" + k: v for k, v in o64.items()})
        rec.update({pre + "f32_grad_" + k: v for k, v in g32.items()})
        rec.update({pre + "f64_grad_" + k: v for k, v in g64.items()})       # kept in float64
        print("wellcond", tag, "reference fp32 vs f64, worst tensor: %.2e" % worst)
    np.savez_compressed(os.path.join(OUT, "wellcond.npz"), **rec)


def gen_preprocess(ref_utils, ref_datasets):
    """The reference's HandDataset.process_single_data (datasets.py:182-403) run on synthetic raw depth frames, with `cv2`
    stubbed by the oracle's restatements of cv2.resize / getRotationMatrix2D / warpAffine / GaussianBlur (cv2 is not installed
    here).  What this pins: everything the reference does AROUND those calls -- crop window, depth threshold, COM centring,
    the shift / scale / rotation augmentation of image and joints, label image, mask, dense targets, normalisation, and the
    un-augmented fallback.  The random draws of the reference (python `random`) are recorded so that the same augmentation can
    be handed to the device pipeline."""
    import random
    from oracle import preprocess_ref as PR, targets_ref as TR
    cv2 = sys.modules["cv2"]
    if getattr(cv2, "__file__", None) is None:              # the empty stand-in of import_reference(): cv2 is absent
        cv2.resize = lambda img, dsize: PR.resize_linear(img, dsize)
        cv2.getRotationMatrix2D = PR.rotation_matrix
        cv2.warpAffine = lambda img, M, dsize: PR.warp_affine(img, M, dsize)
        cv2.GaussianBlur = lambda img, ks, sig: TR.gaussian_blur(np.asarray(img, dtype=np.float64), ks[0], sig)
    else:
        print("gen_preprocess: using the installed OpenCV", cv2.__version__, "-- the fixture is pinned to cv2 itself")
    fx, fy, hu, hv = 588.037, 587.075, 320.0, 240.0          # NYU, datasets.py:693
    H, W, J, S, P = 480, 640, 14, 128, 64

    class Synth(ref_datasets.HandDataset):
        def __init__(self, augment):
            for k, v in dict(fx=fx, fy=fy, halfu=hu, halfv=hv, path="", sigmoid=1.5, image_size=S, kernel_size=7, label_size=P, test_only=False,
                             using_rotation=augment, using_scale=augment, using_shift=augment, using_flip=False, cube_size=150,
                             joint_number=J, config=None, process_mode="uvd", dataset="train", augmentation=augment).items():
                setattr(self, k, v)

        def load_from_text(self, text):
            return text          # the "text" IS the (image, joint_uvd, com, cube) tuple here

        def decode_line_txt(self, text):
            return "synthetic", None     # (only used by the reference to word its error messages, datasets.py:363, 388)

    rng = np.random.default_rng(77)
    rec = {"intrinsics": np.array([fx, fy, hu, hv]), "S": np.int64(S), "P": np.int64(P)}
    n = 6
    frames = []
    for i in range(n):
        # a synthetic depth frame: background 0, a hand-sized blob of smooth depth around the COM, clutter behind it
        cz = 550.0 + 300.0 * rng.random()
        cu, cv_ = 200 + 240 * rng.random(), 150 + 180 * rng.random()
        yy, xx = np.mgrid[0:H, 0:W]
        rad = 150.0 / cz * fx * (0.45 + 0.2 * rng.random())
        blob = ((xx - cu) ** 2 + (yy - cv_) ** 2) < rad ** 2
        depth = np.where(blob, cz + 60 * np.sin(xx / 17.0 + i) * np.cos(yy / 23.0) + 0.1 * (xx - cu), 0.0)
        depth = np.where((xx > cu + 0.5 * rad) & (yy > cv_) & (~blob) & (xx < cu + 1.4 * rad), cz + 400.0, depth)     # far clutter: cut by the cube
        depth = depth.astype(np.float32)
        ang = rng.random(J) * 2 * np.pi
        rr = rad * 0.5 * np.sqrt(rng.random(J))
        juvd = np.stack([cu + rr * np.cos(ang), cv_ + rr * np.sin(ang), cz + 50 * (rng.random(J) - 0.5)], axis=1)
        com = np.array([cu + 3 * (rng.random() - 0.5), cv_ + 3 * (rng.random() - 0.5), cz])
        frames.append((depth, juvd, com, 150))
    # ---- round 3: the reference's edge cases (frames 6..8; the six frames above are unchanged)
    def joint_for(frame, target_uv, a=None):
        """Raw joint position (u, v) whose LABEL-pixel coordinates come out as target_uv on the plain (a is None) or augmented path."""
        c = PR.shift_com(frame[2], a["shift_x"], a["shift_y"]) if a else np.array(frame[2], dtype=np.float64)
        o = PR.process_single(frame[0], frame[1], c, 150, fx, fy, S, P, **({"angle": a["angle"], "scale": a["scale"]} if a else {}))
        cen = (np.asarray(target_uv, dtype=np.float64) - P // 2) / (P - 1) * (S - 1)
        if a:
            ang = a["angle"] / 180.0 * np.pi
            Rot = np.array([[np.cos(ang), np.sin(ang)], [-np.sin(ang), np.cos(ang)]])
            cen = (cen / a["scale"]) @ np.linalg.inv(Rot.T)
        return cen * (o["box_size"] - 1) / (S - 1) + o["com"][:2]
    # frame 6: joint 0 at label pixel (-0.5, -0.4) on the UN-augmented path: numpy wraps heatmap[-1, -1] (utils.py:54-57), no error
    f6 = [a.copy() if hasattr(a, "copy") else a for a in frames[1]]
    f6[1][0, :2] = joint_for(f6, (-0.5, -0.4))
    frames.append(tuple(f6))
    # frame 7: the same on the AUGMENTED path (the draws of seed 1000 + 7 are known in advance): the reference keeps the augmented sample
    f7 = [a.copy() if hasattr(a, "copy") else a for a in frames[2]]
    random.seed(1000 + 7)
    a7 = PR.draws_to_augmentation([random.random() for _ in range(5)])
    f7[1][0, :2] = joint_for(f7, (-0.6, -0.3), a7)
    frames.append(tuple(f7))
    # frame 8: a hand of a few pixels -> sum(mask) < 10 -> ValueError (datasets.py:385-390), both paths
    cz = 700.0
    depth = np.zeros((H, W), np.float32)
    depth[240:243, 320:323] = cz
    juvd = np.stack([320 + 2 * rng.random(J), 240 + 2 * rng.random(J), cz + 5 * (rng.random(J) - 0.5)], axis=1)
    frames.append((depth, juvd, np.array([321.0, 241.0, cz]), 150))
    names = ("img", "label_img", "mask", "box_size", "cube_size", "com", "uvd", "heatmaps", "depthmaps")
    for aug in (False, True):
        ds = Synth(aug)
        for i, fr in enumerate(frames):
            random.seed(1000 + i)
            draws = []
            real = random.random
            random.random = lambda: (draws.append(real()) or draws[-1])
            pre = "%s%d_" % ("aug" if aug else "plain", i)
            import contextlib, io
            try:
                with contextlib.redirect_stdout(io.StringIO()):
                    out = ds.process_single_data((fr[0].copy(), fr[1].copy(), fr[2].copy(), fr[3]))
                rec[pre + "rejected"] = np.array(False)
            except ValueError:
                assert i >= 6, "the round-2 frames are all valid"
                out = None
                rec[pre + "rejected"] = np.array(True)          # the reference raised: check_text drops the sample
            finally:
                random.random = real
            for nm, t in zip(names, out or ()):
                rec[pre + nm] = t.numpy()
            rec[pre + "draws"] = np.array(draws)
        print("preprocess", "augmented" if aug else "plain", "done")
    # what the fixture must contain to be worth anything: the wrap on both paths (kept, not rejected, augmented sample kept), the rejection
    assert not rec["plain6_rejected"] and not rec["aug7_rejected"] and rec["plain8_rejected"] and rec["aug8_rejected"]
    assert not np.array_equal(rec["aug7_img"], rec["plain7_img"]), "frame 7 kept the augmented sample"
    for pre in ("plain6_", "aug7_"):
        h = rec[pre + "heatmaps"][0]
        assert h[0, 0] > 0 and h[-1, -1] > 0 and h[0, -1] > 0 and h[-1, 0] > 0 and h[P // 2, P // 2] == 0, "footprint wrapped to the four corners"
    rec["n_frames"] = np.int64(len(frames))
    for i, fr in enumerate(frames):
        rec["raw%d_depth" % i] = fr[0]
        rec["raw%d_joints" % i] = fr[1]
        rec["raw%d_com" % i] = fr[2]
    np.savez_compressed(os.path.join(OUT, "preprocess.npz"), **rec)
    print("preprocess fixture:", os.path.getsize(os.path.join(OUT, "preprocess.npz")), "bytes")


def gen_checkpoint(ref_model, ref_utils):
    """A checkpoint file written by the REFERENCE's own utils.save_model (utils.py:302-307) from a reference module, plus that
    module's outputs on a fixed batch: the build must load the file with strict=True and reproduce the outputs."""
    from weights_util import fill_state_dict
    from pixelwiseregression_amd.synthetic import make_pose_batch
    kw = dict(stage=2, label_size=16, features=32, level=2, kernel_size=3, norm_method="batch", heatmap_method="softmax")
    torch.manual_seed(4321)
    m = ref_model.PixelwiseRegression(4, **kw)
    sd = fill_state_dict(m.state_dict(), seed=13)
    sd = {k: (v if "num_batches_tracked" not in k else torch.tensor(17)) for k, v in sd.items()}
    m.load_state_dict(sd)
    path = os.path.join(OUT, "reference_checkpoint.pt")
    ref_utils.save_model(m, path, seed=4321, model_param=kw)
    batch = make_pose_batch(2, 4, 32, seed=8)
    m.eval()
    with torch.no_grad():
        res = m(batch["img"], batch["label_img"], batch["mask"])
    rec = {"in_" + k: batch[k].numpy() for k in ("img", "label_img", "mask")}
    for s_, (p_, D_, uvd_) in enumerate(res):
        rec["s%d_p" % s_], rec["s%d_D" % s_], rec["s%d_uvd" % s_] = p_.numpy(), D_.numpy(), uvd_.numpy()
    np.savez_compressed(os.path.join(OUT, "reference_checkpoint_outputs.npz"), **rec)
    print("checkpoint written by the reference:", os.path.getsize(path), "bytes")


def gen_trained(ref_model):
    """BASELINE C2's architecture with TRAINED weights (tools/make_trained_weights.py on the GPU box: 600 AdamW steps of the reference's
    loop on rendered synthetic hands -> gpurun_out/trained_c2_weights.npz) evaluated by the REFERENCE in float64 and in fp32 on four
    held-out frames: the fixture that pins the bf16 (throughput) engine to the reference on a well-conditioned network
    (tests/test_trained_fixture_gpu.py); the CPU oracle is checked against it in tests/test_oracle_golden.py.  Forward + the gradient of
    the train.py:197-205 loss (alpha = 1) in float64."""
    src = os.path.join(ROOT, "gpurun_out", "trained_c2_weights.npz")
    if not os.path.exists(src):       # regenerate from the weights and inputs the committed fixture already holds
        src = os.path.join(OUT, "trained_c2.npz")
    w = dict(np.load(src))
    cfg = dict(joints=14, stage=2, label_size=64, features=128, level=4, kernel_size=3, norm_method="instance", heatmap_method="softmax")
    sd = {k[3:]: torch.from_numpy(w[k]) for k in w if k.startswith("sd_")}
    batch = {k[3:]: torch.from_numpy(w[k]) for k in w if k.startswith("in_")}
    batch["heatmaps"] = torch.zeros(1); batch["depthmaps"] = torch.zeros(1)      # (alpha = 1: the dense terms have weight zero)
    with torch.no_grad():
        out32, _, _ = _run_model(ref_model, cfg, sd, batch, None, train=False)
    out64, _, _ = _run_model(ref_model, cfg, sd, {k: v for k, v in batch.items()}, None, train=False, double=True)
    # Gradient fixture: NOT the training loss -- at a trained point its gradient 2 (uvd - target) / (B J) is as small as the bf16 error of
    # uvd itself, so a bf16 engine's gradient of it is mostly the noise of its own outputs (measured: cosine 0.57 with the float64
    # gradient) -- but a fixed LINEAR functional of all outputs, L = sum_s <uvd_s, GU_s> + <p_s, GH_s> + <D_s, GD_s>, with seeded random
    # weights (GH, GD piecewise constant on 8 x 8 blocks, so that they are small to store): the backward pass of fixed output gradients
    rng = np.random.default_rng(20261003)
    G = {}
    for s_ in range(2):
        G["GU%d" % s_] = (rng.standard_normal((4, 14, 3)) / (4 * 14)).astype(np.float32).astype(np.float64)     # (stored in fp32)
        G["GH%d" % s_] = (rng.standard_normal((4, 14, 8, 8)) * (64.0 / (4 * 14))).astype(np.float32).astype(np.float64)
        G["GD%d" % s_] = (rng.standard_normal((4, 14, 8, 8)) * (0.01 / (4 * 14))).astype(np.float32).astype(np.float64)
    model = ref_model.PixelwiseRegression(cfg["joints"], **{k: v for k, v in cfg.items() if k != "joints"})
    model.load_state_dict(sd, strict=True)
    model = model.double().train()
    res = model(batch["img"].double(), batch["label_img"].double(), batch["mask"].double())
    up = lambda a: torch.from_numpy(np.kron(a, np.ones((8, 8))))
    L = 0
    for s_, (p_, D_, u_) in enumerate(res):
        L = L + (u_ * torch.from_numpy(G["GU%d" % s_])).sum() + (p_ * up(G["GH%d" % s_])).sum() + (D_ * up(G["GD%d" % s_])).sum()
    L.backward()
    g64 = {k: p_.grad.detach().numpy() for k, p_ in model.named_parameters()}
    out64["loss"] = np.float64(L.item())
    rec = {"cfg_" + k: np.array(v) for k, v in cfg.items()}
    rec.update({"sd_" + k: v.numpy() for k, v in sd.items()})
    for k in ("img", "label_img", "mask", "uvd", "box_size", "cube_size", "com"):
        rec["in_" + k] = batch[k].numpy()
    for s in range(2):
        rec["f64_s%d_uvd" % s] = out64["s%d_uvd" % s]
        # maps of the first two frames, stored in fp32 (rounded once from the float64 evaluation); per-map summaries of all four
        rec["f64_s%d_p" % s] = out64["s%d_p" % s][:2].astype(np.float32)
        rec["f64_s%d_D" % s] = out64["s%d_D" % s][:2].astype(np.float32)
        rec["f64_s%d_p_max" % s] = out64["s%d_p" % s].max(axis=(2, 3))
        rec["f64_s%d_p_argmax" % s] = out64["s%d_p" % s].reshape(4, 14, -1).argmax(axis=2)
        rec["f32_s%d_uvd" % s] = out32["s%d_uvd" % s]
    rec["f64_loss"] = out64["loss"]
    for k, v in G.items():
        rec[k] = v.astype(np.float32)
    # the float64 gradient of that functional (training mode), flat in named_parameters order, rounded to bf16 and
    # stored as the upper 16 bits (3 significant digits: enough to bound a bf16 engine's gradient, half the bytes), + per-tensor norms in float64
    gflat = np.concatenate([g64[k].ravel() for k in g64]).astype(np.float32)
    u = gflat.view(np.uint32).astype(np.uint64)
    rec["f64_grad_bf16bits"] = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    rec["f64_grad_norms"] = np.array([float(np.linalg.norm(g64[k].ravel())) for k in g64])
    rec["grad_keys"] = np.array(list(g64.keys()))
    rec["grad_numel"] = np.array([g64[k].size for k in g64])
    # a subset of the tensors in full fp32 (the bf16 image above resolves 2e-3: enough for the bf16 engine, not for the fp32 parity engine):
    # first / last stem conv, a head conv of each stage, the last (J-channel) head conv, a 3x3 of the innermost hourglass level, a stage
    # input conv, the soft-max temperatures, one norm's affine pair
    for k in ("conv.0.weight", "conv.9.weight", "stages.0.plane_regression.conv.3.weight", "stages.1.depth_regression.conv.0.weight",
              "stages.1.depth_regression.conv.9.weight", "stages.0.hourglass.inner.inner.inner.inner.inner.conv.5.weight", "stages.1.conv.weight",
              "stages.0.plane_regression.w", "stages.1.plane_regression.w", "stages.1.plane_regression.conv.4.weight",
              "stages.1.plane_regression.conv.4.bias", "stages.0.hourglass.input_conv.conv.2.weight"):
        rec["f64_grad_" + k] = g64[k].astype(np.float32)
    rec["mm_trained"] = w["mm_trained"]
    rec["train_steps"] = w["train_steps"]
    np.savez_compressed(os.path.join(OUT, "trained_c2.npz"), **rec)
    e = max(float(np.abs(out32["s%d_uvd" % s] - out64["s%d_uvd" % s]).max()) for s in range(2))
    print("trained fixture:", os.path.getsize(os.path.join(OUT, "trained_c2.npz")), "bytes; reference fp32 vs float64 uvd", e)


if __name__ == "__main__":
    if len(sys.argv) > 1:     # (add named fixtures without regenerating the others): targets | wellcond | checkpoint | preprocess | tiny | trained
        os.makedirs(OUT, exist_ok=True)
        torch.set_num_threads(8)
        ref_model, ref_utils, ref_datasets = import_reference()
        for what in sys.argv[1:]:
            {"targets": lambda: gen_targets(ref_utils), "wellcond": lambda: gen_wellcond(ref_model), "tiny": lambda: gen_tiny(ref_model),
             "checkpoint": lambda: gen_checkpoint(ref_model, ref_utils), "trained": lambda: gen_trained(ref_model),
             "preprocess": lambda: gen_preprocess(ref_utils, ref_datasets)}[what]()
        sys.exit(0)
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    ref_model, ref_utils, ref_datasets = import_reference()
    gen_decoder(ref_model)
    gen_tiny(ref_model)
    gen_c1(ref_model)
    gen_metric(ref_utils, ref_datasets)
    gen_init(ref_model)
    gen_targets(ref_utils)
    gen_wellcond(ref_model)
    gen_checkpoint(ref_model, ref_utils)
    gen_preprocess(ref_utils, ref_datasets)
