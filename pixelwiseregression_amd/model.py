"""Drop-in boundary: ``PixelwiseRegression`` with the reference's constructor, forward signature and
state_dict layout (/root/reference/model.py:154-210), computed by the
MI355X-native engine (HIP kernels behind include/pwr.h).

    from pixelwiseregression_amd import PixelwiseRegression
    model = PixelwiseRegression(joints, stage=2, label_size=64, features=128, level=4,
                                norm_method='instance', heatmap_method='softmax').to('cuda')
    results = model(img, label_img, mask)     # list[(heatmaps, depthmaps, uvd)], len = stage

Differences a caller can observe (all additive):
  * parameters are views into one flat fp32 buffer (one fused optimizer / one RCCL all-reduce);
  * ``model.set_precision('bf16' | 'fp32')`` picks the MFMA input type (fp32 = exact f32 MFMA,
    parity mode; bf16 = the training configuration of BASELINE.json); under ``torch.autocast``
    the engine uses bf16;
  * the module refuses to run on the CPU: there is no fallback path.
"""
import math
from collections import OrderedDict

import numpy as np
import torch
from torch import nn


class _Node(nn.Module):
    """Pure namespace: holds Parameters / buffers / child nodes so that state_dict keys equal the
    attribute paths of the reference's module tree.  Never called."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("namespace node")


def com_grid(P):
    """[2,P,P] fp32 expectation grid (utils.py:24-35, model.py:67-71)."""
    ax = (np.arange(P, dtype=np.float64) - (P // 2)) / (P - 1)
    g = np.stack([np.broadcast_to(ax[None, :], (P, P)), np.broadcast_to(ax[:, None], (P, P))])
    return torch.from_numpy(np.ascontiguousarray(g)).float()


class _Builder:
    """Creates parameters in the reference's construction order so that (a) state_dict order and
    names match and (b) the same torch seed yields the same initial weights as the reference."""

    def __init__(self, norm_method):
        if norm_method not in ("batch", "instance"):
            # the reference dies with UnboundLocalError on `norm` (model.py:157-160)
            raise UnboundLocalError("local variable 'norm' referenced before assignment")
        self.norm_method = norm_method
        self.convs = []      # (node) in creation order, for the xavier pass
        self.layers = []     # flat description used by the engine: dicts

    def conv(self, cin, cout, k):
        n = _Node()
        n.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        n.bias = nn.Parameter(torch.empty(cout))
        # torch.nn.Conv2d.reset_parameters
        nn.init.kaiming_uniform_(n.weight, a=math.sqrt(5))
        bound = 1 / math.sqrt(cin * k * k)
        nn.init.uniform_(n.bias, -bound, bound)
        self.convs.append(n)
        return n

    def norm(self, c):
        n = _Node()
        n.weight = nn.Parameter(torch.ones(c))
        n.bias = nn.Parameter(torch.zeros(c))
        if self.norm_method == "batch":
            n.register_buffer("running_mean", torch.zeros(c))
            n.register_buffer("running_var", torch.ones(c))
            n.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        return n

    def seq(self, items):
        """items: list of (index, node)."""
        n = _Node()
        for i, c in items:
            n.add_module(str(i), c)
        return n

    def resblock(self, F, k):
        # model.py:10-20: indices 0 norm, 2 conv1x1, 3 norm, 5 conv kxk, 6 norm, 8 conv1x1
        n = _Node()
        n.conv = self.seq([(0, self.norm(F)), (2, self.conv(F, F // 2, 1)), (3, self.norm(F // 2)),
                           (5, self.conv(F // 2, F // 2, k)), (6, self.norm(F // 2)), (8, self.conv(F // 2, F, 1))])
        return n

    def hourglass(self, F, level, k):
        n = _Node()
        n.input_conv = self.resblock(F, k)
        n.inner = self.hourglass(F, level - 1, k) if level > 0 else self.resblock(F, k)
        n.output_conv = self.resblock(F, k)
        return n

    def head(self, F, J, k):
        return self.seq([(0, self.conv(F, F, k)), (1, self.norm(F)), (3, self.conv(F, F, k)), (4, self.norm(F)),
                         (6, self.conv(F, F, k)), (7, self.norm(F)), (9, self.conv(F, J, k))])

    def xavier(self):
        for n in self.convs:           # utils.py:339-342 via self.apply (model.py:198)
            nn.init.xavier_normal_(n.weight.data)


class PixelwiseRegression(nn.Module):
    def __init__(self, joints, stage=2, label_size=64, features=256, level=4, kernel_size=3, norm_method='batch',
                 heatmap_method='softmax'):
        super().__init__()
        self.joints, self.stage, self.label_size = int(joints), int(stage), int(label_size)
        self.features, self.level, self.kernel_size = int(features), int(level), int(kernel_size)
        self.norm_method, self.heatmap_method = norm_method, heatmap_method
        b = _Builder(norm_method)
        F_, J, k = self.features, self.joints, self.kernel_size
        # ---- stem (model.py:164-187)
        items, idx, c = [(0, b.conv(1, 32, k)), (1, b.norm(32))], 3, 32
        while c < F_:
            nxt = min(2 * c, F_)
            items += [(idx, b.conv(c, nxt, k)), (idx + 1, b.norm(nxt))]
            idx, c = idx + 3, nxt
        items += [(idx, b.conv(F_, F_, k)), (idx + 1, b.norm(F_))]
        self.conv = b.seq(items)
        self.n_stem = len(items) // 2
        # ---- stages (model.py:189-196)
        stages = []
        for s in range(self.stage):
            blk = _Node()
            blk.conv = b.conv(F_ if s == 0 else 2 * J + 1, F_, 1)
            blk.hourglass = b.hourglass(F_, self.level, 3)       # model.py:139: kernel_size not forwarded
            pr = _Node()
            pr.conv = b.head(F_, J, k)
            pr.register_buffer("filter", com_grid(self.label_size))
            if heatmap_method == 'softmax':
                pr.register_parameter("w", nn.Parameter(torch.ones(J, 1)))
            blk.plane_regression = pr
            dr = _Node()
            dr.conv = b.head(F_, J, k)
            blk.depth_regression = dr
            stages.append(blk)
        self.stages = nn.ModuleList(stages)
        b.xavier()
        self._precision = "fp32"       # parity mode by default, like the reference; set_precision("bf16") or torch.autocast selects the bf16 engine
        self._flat = None
        self._flat_grad = None
        self._engine = None
        self._ddp = None
        self._flatten()

    # ------------------------------------------------------------------ flat parameter storage
    def _flatten(self):
        """(Re)pack all parameters into one contiguous fp32 buffer and make them views of it; same for the
        BatchNorm running statistics (fp32) and num_batches_tracked (int64)."""
        params = [p for _, p in self.named_parameters()]
        if not params:
            return
        dev = params[0].device
        total = sum(p.numel() for p in params)
        flat = torch.empty(total, dtype=torch.float32, device=dev)
        off = 0
        self._offsets = OrderedDict()
        with torch.no_grad():
            for name, p in self.named_parameters():
                n = p.numel()
                flat[off:off + n].copy_(p.detach().reshape(-1).float())
                p.data = flat[off:off + n].view(p.shape)
                p.grad = None
                self._offsets[name] = (off, tuple(p.shape))
                off += n
            stats = [(k, b) for k, b in self.named_buffers() if k.endswith("running_mean") or k.endswith("running_var")]
            nbts = [(k, b) for k, b in self.named_buffers() if k.endswith("num_batches_tracked")]
            self._buffer_offsets = []
            self._flat_buf = self._flat_nbt = None
            if stats:
                fb = torch.empty(sum(b.numel() for _, b in stats), dtype=torch.float32, device=dev)
                off = 0
                for k, b in stats:
                    n = b.numel()
                    fb[off:off + n].copy_(b.reshape(-1).float())
                    self._set_buffer(k, fb[off:off + n].view(b.shape))
                    self._buffer_offsets.append(off)
                    off += n
                self._flat_buf = fb
                fn = torch.empty(len(nbts), dtype=torch.long, device=dev)
                for i, (k, b) in enumerate(nbts):
                    fn[i] = b
                    self._set_buffer(k, fn[i])
                self._flat_nbt = fn
        self._flat = flat
        self._flat_grad = None
        self._scratch = None
        self._gviews = None
        self._sviews = None
        self._trig = None
        self._param_list = params
        self._byte_offsets = [4 * o for (o, _) in self._offsets.values()]
        if getattr(self, "_engine", None):
            # plans (and their arenas) are dropped here: their launches may still be in flight on whatever stream ran them
            for d in {p.arena.device for p in self._engine.values() if p.arena.is_cuda}:
                torch.cuda.synchronize(d)
        self._engine = None

    def _set_buffer(self, dotted, tensor):
        mod = self
        parts = dotted.split(".")
        for q in parts[:-1]:
            mod = getattr(mod, q)
        mod._buffers[parts[-1]] = tensor

    def _check_flat(self):
        """Parameters must still alias the flat buffer (deepcopy / manual .data assignment break that)."""
        base = self._flat.data_ptr()
        # EVERY parameter is checked (~340 integer compares, tens of microseconds of host time that the GPU hides): a manual
        # `.data =` or load_state_dict(assign=True) on a middle parameter would otherwise train on stale weights silently
        for p, o in zip(self._param_list, self._byte_offsets):
            if p.data_ptr() != base + o:
                self._flatten()
                return

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._flatten()
        return r

    def __deepcopy__(self, memo):
        new = PixelwiseRegression(self.joints, self.stage, self.label_size, self.features, self.level, self.kernel_size,
                                  self.norm_method, self.heatmap_method)
        new.load_state_dict(self.state_dict())
        new = new.to(self._flat.device)
        new.train(self.training)
        new._precision = self._precision
        for (_, a), (_, b) in zip(self.named_parameters(), new.named_parameters()):
            b.requires_grad_(a.requires_grad)
        return new

    def flat_parameters(self):
        return self._flat

    def flat_grad(self):
        if self._flat_grad is None or self._flat_grad.device != self._flat.device:
            self._flat_grad = torch.zeros_like(self._flat)
            self._gviews = None
        return self._flat_grad

    def _views_of(self, flat):
        return [flat[o:o + int(torch.Size(s).numel())].view(s) for (o, s) in self._offsets.values()]

    def _grad_views(self):
        if self._gviews is None:
            self._gviews = self._views_of(self.flat_grad())
        return self._gviews

    def _grad_scratch(self):
        if self._scratch is None:
            self._scratch = torch.zeros_like(self._flat)
            self._sviews = None
        return self._scratch

    def _scratch_views(self):
        if self._sviews is None:
            self._sviews = self._views_of(self._grad_scratch())
        return self._sviews

    def _trigger(self):
        if self._trig is None or self._trig.device != self._flat.device:
            self._trig = torch.zeros(1, device=self._flat.device, requires_grad=True)
        return self._trig

    def segment_ranges(self):
        """Flat [begin, end) float ranges in backward-completion order: stage S-1, ..., stage 0, stem."""
        names = list(self._offsets.keys())
        def rng(prefix):
            ks = [k for k in names if k.startswith(prefix)]
            o0 = self._offsets[ks[0]][0]
            o1, s1 = self._offsets[ks[-1]]
            return (o0, o1 + int(torch.Size(s1).numel()))
        return [rng("stages.%d." % s) for s in range(self.stage - 1, -1, -1)] + [rng("conv.")]

    def set_precision(self, precision):
        if precision not in ("fp32", "bf16"):
            raise ValueError("precision must be 'fp32' or 'bf16'")
        self._precision = precision
        return self

    def freeze_weight_packs(self, on=True):
        """A promise by the owner that the parameters do not change from here on (until the next call): forwards without gradient then
        re-pack the conv weights (fp32 -> the kernels' bf16 / fp32 fragment order, one launch of ~28 us) once instead of on every call.
        Every call -- True again after the weights were replaced, or False -- ends the previous promise.  Forwards that keep a graph (a plan with gradients) always re-pack."""
        self._pack_count = getattr(self, "_pack_count", 0) + 1
        self._pack_epoch = self._pack_count if on else None
        return self

    # ------------------------------------------------------------------ forward
    def forward(self, img, label_img, mask):
        # error behaviour of the reference's norm layers on a 1x1 innermost map (label_size / 2^(level+1), model.py:25-47):
        # F.instance_norm refuses a single spatial element (in train AND eval: InstanceNorm2d keeps no running statistics),
        # F.batch_norm refuses a single value per channel when training
        inner = self.label_size >> (self.level + 1)
        if inner * inner <= 1:
            if self.norm_method == "instance":
                raise ValueError("Expected more than 1 spatial element when training, got input size torch.Size([%d, %d, %d, %d])"
                                 % (img.shape[0], self.features, inner, inner))
            if self.training and img.shape[0] * inner * inner <= 1:
                raise ValueError("Expected more than 1 value per channel when training, got input size torch.Size([%d, %d, %d, %d])"
                                 % (img.shape[0], self.features, inner, inner))
        if torch.is_grad_enabled() and any(t.requires_grad for t in (img, label_img, mask)):
            # the reference's autograd would differentiate through to the inputs; the engine's backward stops at the parameters
            # (the inputs are data: SURVEY section 8 a-D) -- refuse instead of silently returning no input gradient
            raise NotImplementedError("PixelwiseRegression (MI355X build): gradients with respect to img / label_img / mask are not "
                                      "computed; detach the inputs (they are data in every script of the reference)")
        if not img.is_cuda:
            from ._lib import PwrError
            raise PwrError("PixelwiseRegression (MI355X build) runs on the GPU only; move the module and the "
                           "inputs to 'cuda'. There is no CPU fallback.")
        from .engine import engine_forward
        return engine_forward(self, img, label_img, mask)
