"""Inference throughput with K forward passes in flight on K HIP streams (K independent plans of the same weights, batch 32 each): the
hourglass's small-map launches fill 32 of 256 CUs and every norm costs a launch boundary -- a second batch's kernels run in those holes.
    python tools/infer_streams.py [K ...]      (default 1 2 3)   ->  one JSON line per K: frames/s over all streams, ms per batch per stream
The single-stream figure is bench.py's `infer_frames_per_s`; outputs of stream k are compared with stream 0's (same input, same weights)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pixelwiseregression_amd import PixelwiseRegression
from pixelwiseregression_amd.synthetic import make_batch

dev = torch.device("cuda", 0)
Ks = [int(a) for a in sys.argv[1:]] or [1, 2, 3]
torch.manual_seed(0)
base = PixelwiseRegression(14, stage=2, label_size=64, features=128, level=4, norm_method="instance").to(dev).set_precision("bf16").eval()
b = make_batch(32, 14, S=128, seed=1234, device=dev)
args = (b["img"], b["label_img"], b["mask"])
for K in Ks:
    ms = [base]
    for _ in range(K - 1):
        m = PixelwiseRegression(14, stage=2, label_size=64, features=128, level=4, norm_method="instance").to(dev).set_precision("bf16").eval()
        m.load_state_dict(base.state_dict())
        ms.append(m)
    ss = [torch.cuda.Stream(device=dev) for _ in range(K)]
    outs = [None] * K
    with torch.no_grad():
        def run(n):
            for _ in range(n):
                for k in range(K):
                    with torch.cuda.stream(ss[k]):
                        outs[k] = ms[k](*args)
        run(5)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        N = 100
        run(N)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    same = all(torch.equal(outs[k][-1][2], outs[0][-1][2]) for k in range(1, K))
    print(json.dumps({"streams": K, "frames_per_s": round(K * N * 32 / dt, 1), "ms_per_batch_per_stream": round(dt / N * 1e3, 4),
                      "outputs_equal_across_streams": same}), flush=True)

# the same through serving.StreamedInference (what bench.py reports as infer_frames_per_s_two_streams), from the default stream and from a
# stream of the caller's own
from pixelwiseregression_amd.serving import StreamedInference
for K in Ks:
    for own in (False, True):
        srv = StreamedInference(base, streams=K)
        feed = lambda n: (args for _ in range(n))
        ctx = torch.cuda.stream(torch.cuda.Stream(device=dev)) if own else torch.cuda.stream(torch.cuda.current_stream())
        with ctx:
            for _ in srv.run(feed(6)):
                pass
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in srv.run(feed(200)):
                pass
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print(json.dumps({"StreamedInference": K, "caller": "own stream" if own else "default stream", "frames_per_s": round(200 * 32 / dt, 1)}), flush=True)
        del srv
