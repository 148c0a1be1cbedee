"""Two forward passes in flight (serving.StreamedInference) for N rounds of 2 x 2 batches: every result compared ON THE DEVICE with the
plain single-stream result of its batch (no host synchronisation beyond the helper's own).   python tools/stream_stress.py [rounds = 500]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixelwiseregression_amd import PixelwiseRegression, StreamedInference
from pixelwiseregression_amd.synthetic import make_batch
dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 500
torch.manual_seed(0)
m = PixelwiseRegression(14, stage=2, label_size=64, features=128, level=4, norm_method="instance").to(dev).set_precision("bf16").eval()
bs = [make_batch(32, 14, S=128, seed=40 + i, device=dev) for i in range(4)]
xs = [(b["img"], b["label_img"], b["mask"]) for b in bs]
with torch.no_grad():
    ref = [[tuple(t.clone() for t in st) for st in m(*x)] for x in xs]
srv = StreamedInference(m, streams=2)
bad = torch.zeros((), device=dev)
t0 = time.perf_counter()
for r in range(N):
    for i, out in enumerate(srv.run(iter(xs))):
        for so, sr in zip(out, ref[i]):
            for a, b_ in zip(so, sr):
                bad += (a != b_).any().float()
torch.cuda.synchronize()
print("rounds %d  batches %d  %.2f ms per batch  results that differ from the single-stream pass: %d" % (N, 4 * N, (time.perf_counter() - t0) / (4 * N) * 1e3, int(bad.item())))
