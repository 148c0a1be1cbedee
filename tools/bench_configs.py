"""Train-step / inference throughput of the other BASELINE.json configs on one GPU (per-GPU shard sizes), native harness."""
import os, sys, time, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixelwiseregression_amd import PixelwiseRegression
from pixelwiseregression_amd.synthetic import make_batch
from pixelwiseregression_amd.train import TrainStep
dev = "cuda:0"
cfgs = [("C2 NYU J=14 128x128 B=32", 14, 64, 32), ("C3 MSRA J=21 128x128 B=64", 21, 64, 64), ("C4 HAND17 J=21 128x128 B=32/GPU", 21, 64, 32),
        ("C5 synthetic J=42 256x256 B=128/GPU", 42, 128, 128)]
only = sys.argv[1] if len(sys.argv) > 1 else None
for name, J, P, B in cfgs:
    if only and only not in name: continue
    torch.manual_seed(0)
    m = PixelwiseRegression(J, stage=2, label_size=P, features=128, level=4, norm_method="instance").to(dev).set_precision("bf16").train()
    b = make_batch(B, J, S=2 * P, seed=1, device=dev)
    ts = TrainStep(m, opt="adam", lr=1e-4)
    n = 30 if P == 64 else 6
    for _ in range(3): ts(b["img"], b["label_img"], b["mask"], b["uvd"])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): ts(b["img"], b["label_img"], b["mask"], b["uvd"])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    m.eval()
    with torch.no_grad():
        for _ in range(3): m(b["img"], b["label_img"], b["mask"])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): m(b["img"], b["label_img"], b["mask"])
        torch.cuda.synchronize(); di = (time.perf_counter() - t0) / n
    arena = sum(p.arena.numel() for p in m._engine.values()) / 2**30
    print(json.dumps({"config": name, "train_ms_per_step": dt * 1e3, "train_frames_per_s": B / dt, "infer_frames_per_s": B / di,
                      "loss_finite": bool(torch.isfinite(ts.loss).all()), "arena_GiB": arena, "max_mem_GiB": torch.cuda.max_memory_allocated() / 2**30}))
    del m, ts, b
    torch.cuda.empty_cache()
