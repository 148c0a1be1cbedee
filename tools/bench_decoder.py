"""Decoder micro-benchmark: achieved algorithmic GB/s vs the HBM roofline.

bytes_fwd = 12*B*J*P^2 + 8*B*P^2 + 12*B*J ;  bytes_bwd = 28*B*J*P^2 + 8*B*P^2 (gH and gD given)
"""
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixelwiseregression_amd import ops  # noqa: E402


def run(B, J, P, iters=50):
    dev = torch.device("cuda:0")
    z = torch.randn(B, J, P, P, device=dev)
    D = torch.randn(B, J, P, P, device=dev)
    m = (torch.rand(B, 1, P, P, device=dev) < 0.4).float()
    L = torch.randn(B, 1, P, P, device=dev) * m
    w = torch.ones(J, 1, device=dev)
    gH = torch.randn(B, J, P, P, device=dev)
    gD = torch.randn(B, J, P, P, device=dev)
    gU = torch.randn(B, J, 3, device=dev)
    p, uvd = ops.decode_forward(z, D, L, m, w, "softmax")
    for _ in range(3):
        ops.decode_forward(z, D, L, m, w, "softmax")
        ops.decode_backward(p, z, D, L, m, w, uvd, gH, gD, gU, "softmax")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.decode_forward(z, D, L, m, w, "softmax")
    e1.record(); torch.cuda.synchronize()
    tf = e0.elapsed_time(e1) / iters * 1e-3
    e0.record()
    for _ in range(iters):
        ops.decode_backward(p, z, D, L, m, w, uvd, gH, gD, gU, "softmax")
    e1.record(); torch.cuda.synchronize()
    tb = e0.elapsed_time(e1) / iters * 1e-3
    bf = 12 * B * J * P * P + 8 * B * P * P + 12 * B * J
    bb = 28 * B * J * P * P + 8 * B * P * P
    return {"B": B, "J": J, "P": P, "fwd_us": tf * 1e6, "bwd_us": tb * 1e6, "fwd_GBs": bf / tf / 1e9,
            "bwd_GBs": bb / tb / 1e9, "fwd_frac_8TBs": bf / tf / 8e12, "bwd_frac_8TBs": bb / tb / 8e12}


if __name__ == "__main__":
    for (B, J, P) in ((32, 14, 64), (64, 21, 64), (128, 42, 128)):
        print(json.dumps(run(B, J, P)), flush=True)
