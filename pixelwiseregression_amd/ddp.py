"""Data-parallel training across the GPUs of one node (new capability: the reference is single-GPU,
SURVEY.md section 8e).  One process per GPU; the batch is sharded; every parameter gradient lives in ONE
flat fp32 buffer whose layout follows the backward completion order, so the all-reduce is issued per
backward segment (stage S-1, ..., stage 0, stem) on contiguous slices while the next segment still computes:

    ddp = DataParallel(model)            # broadcasts rank 0's parameters, hooks the engine's backward
    loss.backward()                      # RCCL all-reduce(SUM) of each finished segment overlaps the rest
    optimizer.step()                     # gradients are already averaged

3.3 M parameters = 13 MB fp32: on xGMI this is latency-bound, so three large slices beat per-tensor buckets.
With instance norm samples are independent and the losses are batch means (train.py:197-199), hence the
averaged shard gradients equal the single-GPU big-batch gradient.  With batch norm the statistics stay
per replica (like torch DDP without SyncBatchNorm).
"""
import torch
import torch.distributed as dist


MODES = ("segments", "end")


class DataParallel:
    """mode 'segments' (default): one async all-reduce per finished backward segment, overlapped with the next segment.
    mode 'end': ONE all-reduce of the whole flat gradient after the last segment (no overlap; the yardstick of tools/dist_overhead.py)."""

    def __init__(self, model, process_group=None, broadcast=True, mode="segments"):
        if mode not in MODES:
            raise ValueError("mode must be one of %s" % (MODES,))
        self.mode = mode
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.works = []
        self._comm = None
        self.ranges = model.segment_ranges()
        if broadcast:
            dist.broadcast(model.flat_parameters(), src=0, group=process_group)
            if getattr(model, "_flat_buf", None) is not None:
                dist.broadcast(model._flat_buf, src=0, group=process_group)
        model._ddp = self

    def segment_done(self, model, seg, nseg, plan=None):
        """Called right after backward segment `seg` has been issued.  The engine's parameter-gradient kernels run on side streams that
        the caller's stream joins after the LAST segment only, so the all-reduce of an earlier segment's slice is issued from a
        communication stream that pwr_engine_wait_segment() has made wait for the chain and the side streams: the collective starts when
        that segment's gradients are complete, and the next segment's backward does not stop for it."""
        if self.mode == "end":
            if seg == nseg - 1:
                self.works.append(dist.all_reduce(model.flat_grad(), op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            return
        b, e = self.ranges[seg]
        grad = model.flat_grad()
        if plan is None or not grad.is_cuda or seg == nseg - 1:
            # (the last segment ends with everything joined on the caller's stream; CPU stand-ins of the tests have no streams)
            self.works.append(dist.all_reduce(grad[b:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            return
        from . import _lib
        if self._comm is None:
            self._comm = torch.cuda.Stream(device=grad.device)
        cur = torch.cuda.current_stream(grad.device)
        _lib.check(_lib.lib().pwr_engine_wait_segment(plan.h, cur.cuda_stream, self._comm.cuda_stream), "pwr_engine_wait_segment")
        with torch.cuda.stream(self._comm):
            self.works.append(dist.all_reduce(grad[b:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def wait(self):
        """Block the current stream until every segment's all-reduce (SUM) has finished; the caller scales by 1/world."""
        for w in self.works:
            w.wait()
        self.works = []

    def finish(self, model):
        self.wait()
        model.flat_grad().mul_(1.0 / self.world)
