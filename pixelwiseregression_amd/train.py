"""Native train step: the inner loop of /root/reference/train.py:158-212 without autograd or torch.optim.

    step = TrainStep(model, opt='adam', lr=1e-3, alpha=1.0)        # defaults of train.py:49-57
    loss = step(img, label_img, mask, uvd, heatmaps, depthmaps)     # one optimizer step, returns the loss tensor
    step.epoch_end()                                                # StepLR(step_size=decay_epoch, gamma=lr_decay), train.py:143,212

Forward = one engine call, loss + output gradients = `pwr_loss_sqdiff` (train.py:195-205; with alpha == 1 the dense
heat-map / depth-map terms have weight zero and are skipped instead of being multiplied by 0), backward = one engine call per
segment (with the RCCL all-reduce of ddp.DataParallel in between), update = `pwr_adamw_step` / `pwr_sgd_step` on the flat
parameter buffer (torch.optim.AdamW(weight_decay=0) / SGD(momentum=beta1) semantics, train.py:139-142).
"""
import ctypes

import torch

from . import _lib
from .engine import _get_plan, _run_forward, BF16, F32


class TrainStep:
    def __init__(self, model, opt="adam", lr=1e-3, beta1=0.9, beta2=0.999, weight_decay=0.0, eps=1e-8, alpha=1.0, lambda_h=1.0,
                 lambda_d=0.01, decay_epoch=15, lr_decay=0.2, target_kernel_size=7, target_sigmoid=1.5):
        if opt not in ("adam", "sgd"):
            raise ValueError("opt must be 'adam' or 'sgd' (train.py:139-142)")
        self.model, self.opt = model, opt
        self.lr, self.beta1, self.beta2, self.wd, self.eps = lr, beta1, beta2, weight_decay, eps
        self.alpha, self.lambda_h, self.lambda_d = alpha, lambda_h, lambda_d
        self.decay_epoch, self.lr_decay = decay_epoch, lr_decay
        self.target_kernel_size, self.target_sigmoid = target_kernel_size, target_sigmoid   # train.py:29-30
        self.steps, self.epochs = 0, 0
        flat = model.flat_parameters()
        self.m = torch.zeros_like(flat)
        self.v = torch.zeros_like(flat) if opt == "adam" else None
        self.loss = torch.zeros(1, device=flat.device)
        self._scratch = None

    def epoch_end(self):
        """StepLR: lr *= gamma every `decay_epoch` epochs."""
        self.epochs += 1
        if self.epochs % int(self.decay_epoch) == 0:
            self.lr *= self.lr_decay

    def __call__(self, img, label_img, mask, uvd, heatmaps=None, depthmaps=None):
        m = self.model
        if not m.training:
            raise _lib.PwrError("TrainStep needs model.train()")
        l = _lib.lib()
        dev = img.device
        B, J, P = img.shape[0], m.joints, m.label_size
        dtype = BF16 if (m._precision == "bf16" or torch.is_autocast_enabled()) else F32
        m._check_flat()
        flat = m.flat_parameters()
        if self.m.device != flat.device or self.m.numel() != flat.numel():
            # model.to(...) / a re-flatten moved the parameters: optimizer state on the old buffer would be applied to the
            # wrong device or the wrong elements
            raise _lib.PwrError("TrainStep: the module's flat parameter buffer changed (device %s -> %s, %d -> %d elements) after the "
                                "optimizer state was created; build a new TrainStep after model.to(...) / load of another architecture"
                                % (self.m.device, flat.device, self.m.numel(), flat.numel()))
        img, label_img, mask, uvd = (t.contiguous().float() for t in (img, label_img, mask, uvd))
        plan = _get_plan(m, B, dtype, True)
        outs = _run_forward(m, plan, img, label_img, mask)
        stream = _lib.stream_ptr(dev)
        dense = self.alpha != 1.0
        if dense and (heatmaps is None or depthmaps is None):
            # the dense targets of train.py:197-198, built on the device from the joints (datasets.py:285-294, :365-383)
            from .targets import make_targets
            heatmaps, depthmaps = make_targets(uvd, label_img, mask, self.target_kernel_size, self.target_sigmoid)
        n_map, n_uvd = B * J * P * P, B * J * 3
        if self._scratch is None or self._scratch[0] != (B, dense):
            nb = l.pwr_loss_blocks(n_map if dense else n_uvd)
            bufs = {"partial": torch.empty(nb, device=dev)}
            for s in range(m.stage):
                bufs["gu%d" % s] = torch.empty(B, J, 3, device=dev)
                if dense:
                    bufs["gh%d" % s] = torch.empty(B, J, P, P, device=dev)
                    bufs["gd%d" % s] = torch.empty(B, J, P, P, device=dev)
            self._scratch = ((B, dense), bufs)
        bufs = self._scratch[1]
        ptrs = []
        first = True
        for s in range(m.stage):
            p, D, u = outs[3 * s], outs[3 * s + 1], outs[3 * s + 2]
            gh = gd = None
            if dense:
                gh, gd = bufs["gh%d" % s], bufs["gd%d" % s]
                heat_t, dep_t = heatmaps.contiguous().float(), depthmaps.contiguous().float()
                sc = (1.0 - self.alpha) * self.lambda_h / (B * J)
                _lib.check(l.pwr_loss_sqdiff(p.data_ptr(), heat_t.data_ptr(), gh.data_ptr(), sc, bufs["partial"].data_ptr(),
                                             self.loss.data_ptr(), 0 if first else 1, n_map, stream), "pwr_loss_sqdiff")
                first = False
                sc = (1.0 - self.alpha) * self.lambda_d / (B * J)
                _lib.check(l.pwr_loss_sqdiff(D.data_ptr(), dep_t.data_ptr(), gd.data_ptr(), sc, bufs["partial"].data_ptr(),
                                             self.loss.data_ptr(), 1, n_map, stream), "pwr_loss_sqdiff")
            gu = bufs["gu%d" % s]
            _lib.check(l.pwr_loss_sqdiff(u.data_ptr(), uvd.data_ptr(), gu.data_ptr(), self.alpha / (B * J), bufs["partial"].data_ptr(),
                                         self.loss.data_ptr(), 0 if first else 1, n_uvd, stream), "pwr_loss_sqdiff")
            first = False
            ptrs += [gh.data_ptr() if gh is not None else None, gd.data_ptr() if gd is not None else None, gu.data_ptr()]
        arr = (ctypes.c_void_p * len(ptrs))(*ptrs)
        grad = m.flat_grad()
        plan.bind(m, grad)
        n = grad.numel()
        ddp = m._ddp
        for seg in range(plan.n_seg):
            _lib.check(l.pwr_engine_backward(plan.h, arr, seg, n, stream), "pwr_engine_backward")
            if ddp is not None:
                ddp.segment_done(m, seg, plan.n_seg, plan)
        scale = 1.0
        if ddp is not None:
            ddp.wait()
            scale = 1.0 / ddp.world
        self.steps += 1
        if self.opt == "adam":
            _lib.check(l.pwr_adamw_step(flat.data_ptr(), grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), n, self.lr, self.beta1,
                                        self.beta2, self.eps, self.wd, self.steps, scale, stream), "pwr_adamw_step")
        else:
            _lib.check(l.pwr_sgd_step(flat.data_ptr(), grad.data_ptr(), self.m.data_ptr(), n, self.lr, self.beta1, self.wd,
                                      1 if self.steps == 1 else 0, scale, stream), "pwr_sgd_step")
        self._keep = (outs, img, label_img, mask, uvd)      # alive until the next step: the stream may still read them
        return self.loss.clone()      # a fresh tensor per step (self.loss is overwritten by the next step)
