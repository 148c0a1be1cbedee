"""What does the vendor GEMM (hipBLASLt through torch.matmul, bf16) reach on this GPU at the shape of the dominant conv seen as an
implicit GEMM (M = B*64*64 = 131072 pixels, N = 128 output channels, K = 9*128 = 1152) -- a calibration of what fraction of the
2.5 PFLOP/s peak a well-tuned kernel gets at N = 128, not a product path."""
import json, torch
dev = "cuda:0"
def t(M, N, K, iters=50):
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16); b = torch.randn(K, N, device=dev, dtype=torch.bfloat16)
    for _ in range(5): a @ b
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): a @ b
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    print(json.dumps({"M": M, "N": N, "K": K, "us": round(us, 2), "TFLOPs": round(2.0 * M * N * K / us / 1e6, 1), "frac_of_2500": round(2.0 * M * N * K / us / 1e6 / 2500, 3)}))
t(131072, 128, 1152)
t(131072, 128, 128)
t(131072, 256, 1152)
t(16384, 1024, 1152)
t(8192, 8192, 8192, 10)
# wgrad seen as a GEMM: M = 1152 (9 taps x 128 ci), N = 128 co, K = 131072 pixels
a = torch.randn(131072, 1152, device=dev, dtype=torch.bfloat16); b = torch.randn(131072, 128, device=dev, dtype=torch.bfloat16)
for _ in range(5): a.t() @ b
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): a.t() @ b
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 50 * 1e3
print(json.dumps({"wgrad-shaped A^T B": True, "M": 1152, "N": 128, "K": 131072, "us": round(us, 2), "TFLOPs": round(2.0 * 1152 * 128 * 131072 / us / 1e6, 1)}))
