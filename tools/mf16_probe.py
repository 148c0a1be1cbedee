"""A/B of the MFMA shape in the dominant conv (debug build: PWR_PATCH_MF16=1 selects v_mfma_f32_16x16x32_bf16): correctness against
F.conv2d in float64 on a small batch, then timing at the C2 head shape on random operands, interleaved in one process is impossible
(the switch is read once), so run it twice on the same box."""
import sys, os, json
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dbglib  # noqa: F401
from pixelwiseregression_amd import kernels as K
dev = "cuda:0"
torch.manual_seed(0)
Bs, P, C = 3, 64, 128
x = torch.randn(Bs, P, P, C, device=dev).to(torch.bfloat16)
w = torch.randn(C, C, 3, 3, device=dev) * 0.03
bias = torch.randn(C, device=dev) * 0.1
pack = K.pack_conv(w, 0, K.BF16)
y, _ = K.conv_fwd(x, pack, C, 3, 1, bias=bias)
ref = F.conv2d(x.double().cpu().permute(0, 3, 1, 2), w.to(torch.bfloat16).double().cpu(), bias.double().cpu(), padding=1).permute(0, 2, 3, 1)
err = (y.double().cpu() - ref).abs().max().item() / ref.abs().max().item()
B = 32
x = torch.randn(B, P, P, C, device=dev).to(torch.bfloat16)
st = K.norm_stats(x, torch.ones(C, device=dev), torch.zeros(C, device=dev), mode=0)
def timeit(fn, iters=300):
    for _ in range(30): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
print(json.dumps({"PWR_PATCH_MF16": os.environ.get("PWR_PATCH_MF16", "0"), "max_rel_err_vs_f64": err,
                  "us_fwd_with_norm": round(timeit(lambda: K.conv_fwd(x, pack, C, 3, 1, bias=bias, norm=st)), 2),
                  "us_dgrad_form": round(timeit(lambda: K.conv_fwd(x, pack, C, 3, 1)), 2)}))
