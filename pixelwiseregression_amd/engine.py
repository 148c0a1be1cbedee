"""Python side of the network engine (csrc/engine.hip): plan cache, buffers, autograd glue.

One ``_Plan`` per (batch, dtype, grad-enabled) holds the C++ launch plan, its activation arena and the
packed weights.  A forward is ONE ctypes call; a backward is one call per segment (stage S-1 .. stage 0,
stem) so that the flat gradient of a finished segment can be all-reduced (RCCL) while the next segment
computes (ddp.py).
"""
import ctypes

import torch

from . import _lib

F32, BF16 = 0, 1
# Side streams (parameter-gradient kernels) are joined into the caller's stream after the LAST backward segment (round 4; a join per segment is a
# debug-build experiment, include/pwr_debug.h).


class _Plan:
    def __init__(self, model, B, dtype, need_grad):
        l = _lib.lib()
        self.B, self.dtype, self.need_grad = B, dtype, need_grad
        dev = model._flat.device
        cfg = (ctypes.c_int * 8)(model.joints, model.stage, model.label_size, model.features, model.level, model.kernel_size,
                                 0 if model.norm_method == "instance" else 1, 0 if model.heatmap_method == "softmax" else 1)
        offs = [o for (o, _) in model._offsets.values()]
        nums = [int(torch.Size(s).numel()) for (_, s) in model._offsets.values()]
        n = len(offs)
        poff = (ctypes.c_longlong * n)(*offs)
        pnum = (ctypes.c_longlong * n)(*nums)
        boffs = model._buffer_offsets
        nb = len(boffs)
        boff = (ctypes.c_longlong * max(nb, 1))(*boffs) if nb else None
        self.h = l.pwr_engine_create(cfg, B, dtype, 1 if need_grad else 0, poff, pnum, n, boff, nb)
        if not self.h:
            raise _lib.PwrError("pwr_engine_create failed: %s" % (l.pwr_last_error() or b"").decode())
        self.arena = torch.empty(l.pwr_engine_arena_bytes(self.h), dtype=torch.uint8, device=dev)
        self.packs = torch.empty(l.pwr_engine_pack_bytes(self.h), dtype=torch.uint8, device=dev)
        nd = l.pwr_engine_desc_bytes(self.h)
        host = (ctypes.c_char * nd)()
        l.pwr_engine_get_descs(self.h, host)
        off = l.pwr_engine_desc_offset(self.h)
        self.packs[off:off + nd].copy_(torch.frombuffer(bytearray(host.raw), dtype=torch.uint8))
        self.n_seg = l.pwr_engine_num_segments(self.h)
        self.packed_version = None
        self.bound = None
        self.outs_t = (ctypes.c_void_p * (3 * model.stage))

    def bind(self, model, grads):
        key = (model._flat.data_ptr(), grads.data_ptr() if grads is not None else 0,
               model._flat_buf.data_ptr() if model._flat_buf is not None else 0)
        if key != self.bound:
            _lib.check(_lib.lib().pwr_engine_bind(self.h, self.arena.data_ptr(), self.packs.data_ptr(), key[0], key[1] or None,
                                                  key[2] or None), "pwr_engine_bind")
            self.bound = key
            self.packed_version = None

    def __del__(self):
        try:
            if self.h:
                _lib.lib().pwr_engine_destroy(self.h)
        except Exception:
            pass


MAX_PLANS = 4          # per module: e.g. (train B), (eval B), (ragged last eval batch), one spare


PLAN_BUDGET_BYTES = None      # arena bytes the cached plans of one module may hold together; None = half of the device's memory


def _plan_budget_bytes(dev):
    """Arena bytes the cached plans of one module may hold together (least recently used plans beyond it are dropped):
    engine.PLAN_BUDGET_BYTES, default half of the device's memory.  (A module attribute, not an environment variable: the package, like
    the library, reads no PWR_* variable.)"""
    if PLAN_BUDGET_BYTES is not None:
        return int(PLAN_BUDGET_BYTES)
    try:
        return torch.cuda.get_device_properties(dev).total_memory // 2
    except Exception:
        return 64 << 30


def _get_plan(model, B, dtype, need_grad):
    """The plan of (batch, dtype, grad) for this module, most recently used last.  A plan owns an activation arena of
    2-54 GB, so the cache is bounded: at most MAX_PLANS plans and _plan_budget_bytes() of arena; older ones are evicted
    (a ragged last validation batch no longer doubles the footprint for the rest of the run)."""
    if model._engine is None:
        model._engine = {}
    cache = model._engine
    key = (B, dtype, need_grad)
    plan = cache.pop(key, None)
    if plan is None:
        plan = _Plan(model, B, dtype, need_grad)
    cache[key] = plan                      # dicts keep insertion order: re-inserting makes it the most recent
    budget = _plan_budget_bytes(model._flat.device)
    while len(cache) > 1 and (len(cache) > MAX_PLANS or sum(p.arena.numel() for p in cache.values()) > budget):
        old_key = next(iter(cache))
        if old_key == key:
            break
        old = cache.pop(old_key)
        torch.cuda.synchronize(model._flat.device)    # its launches may still be in flight, on ANY stream it was last run on (rare path)
        del old
    return plan


def _run_forward(model, plan, img, label_img, mask):
    l = _lib.lib()
    dev = img.device
    B, J, P = plan.B, model.joints, model.label_size
    plan.bind(model, model.flat_grad() if plan.need_grad else None)
    stream = _lib.stream_ptr(dev)
    # Re-pack the conv weights from the flat fp32 parameters on every forward (one launch, ~10 us): in-place updates
    # through the per-parameter views (optimizers, load_state_dict) do not bump the flat buffer's version counter,
    # so there is no cheap, reliable "unchanged" test -- and a stale pack would be silently wrong.
    # (round 6: the pack is issued INSIDE the forward call, on a side stream beside the stem's first conv)
    # The one exception: a module whose owner PROMISED that its parameters stand still (model.freeze_weight_packs(), taken by
    # serving.StreamedInference for its private replicas): a no-grad plan then packs once per promise.
    epoch = getattr(model, "_pack_epoch", None)
    if plan.need_grad or epoch is None or plan.packed_version != epoch:
        _lib.check(l.pwr_engine_pack_beside_forward(plan.h), "pwr_engine_pack_beside_forward")
        plan.packed_version = epoch
    outs = []
    for _ in range(model.stage):
        outs += [torch.empty(B, J, P, P, device=dev, dtype=torch.float32), torch.empty(B, J, P, P, device=dev, dtype=torch.float32),
                 torch.empty(B, J, 3, device=dev, dtype=torch.float32)]
    arr = plan.outs_t(*[t.data_ptr() for t in outs])
    training = 1 if model.training else 0
    _lib.check(l.pwr_engine_forward(plan.h, img.data_ptr(), label_img.data_ptr(), mask.data_ptr(), arr, training, stream),
               "pwr_engine_forward")
    if training and model._flat_nbt is not None:
        model._flat_nbt += 1
    return outs


class _EngineFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, trigger, model, plan, img, label_img, mask):
        outs = _run_forward(model, plan, img, label_img, mask)
        ctx.model, ctx.plan = model, plan
        ctx.generation = _lib.lib().pwr_engine_generation(plan.h)
        # The backward re-reads the inputs and the outputs.  They go through save_for_backward, NOT onto ctx as attributes:
        # `ctx.outs = outs` would form the cycle output -> grad_fn -> ctx -> output, which the garbage collector cannot
        # break (six outputs share one grad_fn), and every training forward would leak its output maps.
        ctx.save_for_backward(img, label_img, mask, *outs)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        model, plan = ctx.model, ctx.plan
        l = _lib.lib()
        if l.pwr_engine_generation(plan.h) != ctx.generation:
            raise _lib.PwrError("backward() after another forward() of the same (batch, dtype) plan: the engine keeps one "
                                "set of activations per plan. Call backward before the next training forward.")
        saved = ctx.saved_tensors
        outs = saved[3:]
        dev = outs[0].device
        stream = _lib.stream_ptr(dev)
        keep = []
        ptrs = []
        for g, o in zip(gouts, outs):
            if g is None:
                ptrs.append(None)
            else:
                g = g.contiguous().float()
                keep.append(g)
                ptrs.append(g.data_ptr())
        arr = (ctypes.c_void_p * len(ptrs))(*ptrs)
        # where do the gradients go?  Fresh (p.grad is None everywhere) -> straight into the flat buffer.
        flat_grad = model.flat_grad()
        views = model._grad_views()
        params = model._param_list
        fresh = all(p.grad is None for p in params)
        target = flat_grad if fresh else model._grad_scratch()
        plan.bind(model, target)
        n = flat_grad.numel()
        ddp = model._ddp
        for seg in range(plan.n_seg):
            _lib.check(l.pwr_engine_backward(plan.h, arr, seg, n, stream), "pwr_engine_backward")
            if ddp is not None and fresh:
                ddp.segment_done(model, seg, plan.n_seg, plan)
        if fresh:
            for p, v in zip(params, views):
                if p.requires_grad:
                    p.grad = v
            if ddp is not None:
                ddp.finish(model)
        else:
            sv = model._scratch_views()
            for p, v, s in zip(params, views, sv):
                if not p.requires_grad:
                    continue
                if p.grad is None:
                    v.copy_(s)
                    p.grad = v
                else:
                    p.grad.add_(s)
            if ddp is not None:
                raise _lib.PwrError("gradient accumulation under the built-in data-parallel mode is not supported")
        return (None,) * 6


_warned_fp16 = False


def _warn_fp16_autocast():
    """/root/reference/train.py:170-189 runs `torch.cuda.amp.autocast()` -- float16 -- with a GradScaler.  This engine has ONE reduced
    precision, bfloat16 (BASELINE config: "Adam, bf16"): same 16-bit storage and the same MFMA rate on gfx950, 8 exponent bits instead of
    5, so nothing overflows and the caller's GradScaler never has to skip a step.  The substitution is not silent: say so, once."""
    global _warned_fp16
    if not _warned_fp16:
        _warned_fp16 = True
        import warnings
        warnings.warn("pixelwiseregression_amd: float16 autocast requested; the engine computes convolutions in bfloat16 instead "
                      "(fp32 accumulation, fp32 decoder).  A GradScaler around it is harmless and never sees an overflow.  Use "
                      "torch.autocast('cuda', dtype=torch.bfloat16) or model.set_precision('bf16') to silence this.", UserWarning, stacklevel=3)


def engine_forward(model, img, label_img, mask):
    B = img.shape[0]
    S, P = 2 * model.label_size, model.label_size
    if tuple(img.shape) != (B, 1, S, S) or tuple(label_img.shape) != (B, 1, P, P) or tuple(mask.shape) != (B, 1, P, P):
        # the reference fails at .view(1,1,H,W) (model.py:92) when image side != 2*label_size
        raise RuntimeError("expected img [B,1,%d,%d], label_img/mask [B,1,%d,%d]; got %s %s %s" %
                           (S, S, P, P, tuple(img.shape), tuple(label_img.shape), tuple(mask.shape)))
    model._check_flat()
    img, label_img, mask = (t.contiguous().float() for t in (img, label_img, mask))
    autocast = torch.is_autocast_enabled()
    if autocast and torch.get_autocast_dtype("cuda") == torch.float16:
        _warn_fp16_autocast()
    dtype = BF16 if (model._precision == "bf16" or autocast) else F32
    need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in model._param_list)
    plan = _get_plan(model, B, dtype, need_grad)
    if need_grad:
        outs = _EngineFn.apply(model._trigger(), model, plan, img, label_img, mask)
    else:
        outs = _run_forward(model, plan, img, label_img, mask)
    return [(outs[3 * s], outs[3 * s + 1], outs[3 * s + 2]) for s in range(model.stage)]
