"""Validation pass of the reference's training loop (/root/reference/train.py:230-285) on the engine: forward under
``no_grad`` + ``eval()``, the three loss terms per stage, and the mean 3D joint error in millimetres
(``recover_uvd`` -> ``uvd2xyz`` -> mean over joints of the Euclidean distance, train.py:271-276,285).

    errs, losses = validate(model, batches, dataset="NYU")       # errs[s] = mean mm error of stage s over all samples
    errs, losses = validate(model, batches, dataset="NYU", streams=2)      # the same numbers, two batches in flight (serving.py)

``batches`` is an iterable of dicts with the reference loader's fields (img, label_img, mask, box_size, cube_size, com,
uvd[, heatmaps, depthmaps]); the dense loss terms are skipped when the dense targets are absent (alpha == 1 training).

``train_and_validate`` is the build's stand-in for "run train.py for a while": the reference's loop (train.py:158-212) on a
stream of rendered synthetic hands (synthetic.make_pose_batch, a new seed per step), validated on held-out batches.
"""
import collections

import numpy as np
import torch

from .metric import INTRINSICS, recover_uvd, uvd2xyz, mean_joint_error


def _forwards(model, batches, streams):
    """(batch, results) in the order of the batches; streams > 1: through serving.StreamedInference (the helper and its plans are kept on
    the module between calls; its replicas get the module's current weights at every call), so that the host's metric arithmetic and
    the .cpu() copies of one batch run beside the next batch's forward pass."""
    if streams <= 1 or not next(model.parameters()).is_cuda:
        for b in batches:
            yield b, model(b["img"], b["label_img"], b["mask"])
        return
    from .serving import StreamedInference
    srv = model.__dict__.get("_streamed")
    if srv is None or len(srv.streams) != streams:
        srv = StreamedInference(model, streams)
        model.__dict__["_streamed"] = srv
    else:
        srv.refresh()
    waiting = collections.deque()

    def feed():
        for b in batches:
            waiting.append(b)
            yield b["img"], b["label_img"], b["mask"]
    for results in srv.run(feed()):
        yield waiting.popleft(), results


def validate(model, batches, dataset="NYU", alpha=1.0, lambda_h=1.0, lambda_d=0.01, streams=1):
    intr = INTRINSICS[dataset]
    was_training = model.training
    model.eval()
    per_stage, losses, num = None, None, 0
    with torch.no_grad():
        for b, results in _forwards(model, batches, streams):
            num += 1
            if per_stage is None:
                per_stage = [[] for _ in results]
                losses = [[0.0, 0.0, 0.0] for _ in results]
            box, cube, com = b["box_size"].cpu(), b["cube_size"].cpu(), b["com"].cpu()
            true_xyz = uvd2xyz(recover_uvd(b["uvd"].float().cpu().clone(), box, com, cube).numpy(), *intr)
            for i, (heat, depth, uvd) in enumerate(results):
                if "heatmaps" in b and "depthmaps" in b:
                    losses[i][0] += float(lambda_h * torch.mean(torch.sum((heat - b["heatmaps"]) ** 2, dim=(2, 3))))
                    losses[i][1] += float(lambda_d * torch.mean(torch.sum((depth - b["depthmaps"]) ** 2, dim=(2, 3))))
                losses[i][2] += float(torch.mean(torch.sum((uvd - b["uvd"]) ** 2, dim=2)))
                xyz = uvd2xyz(recover_uvd(uvd.float().cpu().clone(), box, com, cube).numpy(), *intr)
                per_stage[i].append(mean_joint_error(xyz, true_xyz))
    model.train(was_training)
    errs = [float(np.mean(np.concatenate(e, axis=0))) for e in per_stage]
    losses = [[v / num for v in l] for l in losses]
    total = sum(alpha * l[2] + (1 - alpha) * (l[0] + l[1]) for l in losses)
    return errs, {"per_stage": losses, "val_loss": total}


def train_and_validate(precision, steps, B=32, J=14, S=128, lr=1e-3, eval_every=100, n_val=4, seed=0, dev="cuda:0", features=128, level=4,
        stage=2, log=None, opt="adam"):
    import time
    from .model import PixelwiseRegression
    from .synthetic import make_pose_batch
    from .train import TrainStep
    torch.manual_seed(seed)
    m = PixelwiseRegression(J, stage=stage, label_size=S // 2, features=features, level=level, norm_method="instance")
    m = m.to(dev).set_precision(precision).train()
    ts = TrainStep(m, opt=opt, lr=lr, alpha=1.0)
    val = [make_pose_batch(B, J, S, seed=10_000_000 + k, device=dev) for k in range(n_val)]
    curve = []
    errs, _ = validate(m, val)
    curve.append({"step": 0, "mm": errs})
    losses = []
    t0 = time.time()
    for it in range(steps):
        b = make_pose_batch(B, J, S, seed=seed * 1_000_003 + it + 1, device=dev)
        loss = ts(b["img"], b["label_img"], b["mask"], b["uvd"])
        losses.append(loss)
        if (it + 1) % eval_every == 0 or it + 1 == steps:
            errs, vl = validate(m, val)
            curve.append({"step": it + 1, "mm": errs, "val_loss": vl["val_loss"], "train_loss": float(torch.stack(losses[-eval_every:]).mean())})
            if log:
                log("%s step %d  train loss %.4e  val loss %.4e  mm per stage %s" % (precision, it + 1, curve[-1]["train_loss"], vl["val_loss"],
                                                                                    ["%.2f" % e for e in errs]))
    torch.cuda.synchronize()
    return {"precision": precision, "steps": steps, "batch": B, "lr": lr, "opt": opt, "seconds": time.time() - t0,
            "train_loss": [float(x) for x in torch.stack(losses).flatten().cpu()], "curve": curve, "final_mm": curve[-1]["mm"][-1]}
