"""Evaluation loop of the reference (/root/reference/test.py:89-102: ``load_model(..., eval_mode=True)``, ``with torch.no_grad()``,
``for batch in iter(test_loader): results = model(img, label_img, mask)``) with K batches in flight on K HIP streams.

Why: one forward pass is a chain of ~95 dependent launches, a third of them on maps of 16 x 16 pixels and less (32 workgroups on a
256-CU chip) or a norm's finalisation (a launch boundary each).  A second batch's full-chip convolutions run in those holes: two
plans (two arenas, the same weights) on two streams measured 27.8 k frames/s against 21.4 k for one at the BASELINE C2 shape; a
third adds nothing (the weight-stationary convs take a whole CU each, and one Python thread issues ~0.4 ms of launches per pass).
Results are the single-stream results bit for bit: every plan runs the same kernels on its own buffers.
"""
import copy
import collections

import torch


class StreamedInference:
    """``for out in StreamedInference(model, streams=2).run(batches): ...`` -- ``batches`` yields ``(img, label_img, mask)`` device
    tensors; ``out`` is what ``model(img, label_img, mask)`` returns (a list of ``(heatmaps, depthmaps, uvd)`` per stage), in the
    order of the batches, complete (host-synchronised) when it is handed out.

    Every stream runs a PRIVATE deep copy of ``model`` taken at construction (``refresh()`` copies the weights again after the model
    was trained further).  Nobody else can touch a copy's parameters, so the copies re-pack their conv weights once per refresh instead
    of on every forward (``freeze_weight_packs``: one 28-us launch less per pass)."""

    def __init__(self, model, streams=2):
        if streams < 1:
            raise ValueError("streams must be >= 1")
        if not next(model.parameters()).is_cuda:
            raise ValueError("StreamedInference needs the model on a HIP device")
        self.model = model
        self.replicas = [copy.deepcopy(model) for _ in range(streams)]
        for m in self.replicas:
            m.eval().freeze_weight_packs(True)
        dev = next(model.parameters()).device
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(streams)]

    def refresh(self):
        sd = self.model.state_dict()
        for m in self.replicas:
            m.load_state_dict(sd)
            m.freeze_weight_packs(True)

    def run(self, batches):
        """Order between the caller's stream and the K worker streams is kept by the HOST, not by stream-to-stream waits: the runtime
        multiplexes a process's streams onto a few hardware queues (four by default), and a wait that lands in a queue shared with the
        OTHER worker stream holds that worker back until the awaited batch is done -- the two passes then run one after the other
        (measured: 16 k instead of 27 k frames/s whenever the waiting stream happened to share a queue).  So: the caller's stream is
        synchronised before a batch is submitted (the batch may have been produced on it), and a result is handed out after its
        event has completed on the host; K - 1 passes stay in flight meanwhile."""
        caller = torch.cuda.current_stream()
        pending = collections.deque()
        K = len(self.streams)
        with torch.no_grad():
            for i, (img, label_img, mask) in enumerate(batches):
                k = i % K
                s = self.streams[k]
                caller.synchronize()
                with torch.cuda.stream(s):
                    out = self.replicas[k](img, label_img, mask)
                    ev = torch.cuda.Event()
                    ev.record(s)
                for t in (img, label_img, mask):
                    t.record_stream(s)
                pending.append((out, ev))
                if len(pending) == K:
                    yield self._hand_over(pending.popleft(), caller)
            while pending:
                yield self._hand_over(pending.popleft(), caller)

    @staticmethod
    def _hand_over(item, caller):
        out, ev = item
        ev.synchronize()
        for stage in out:
            for t in stage:
                t.record_stream(caller)               # allocated on the worker's stream, consumed on the caller's
        return out
