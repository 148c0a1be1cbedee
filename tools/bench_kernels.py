"""Isolated timing of the two hot kernels at the BASELINE C2 heads shape (for rocprofv3 --pmc runs)."""
import sys, os, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixelwiseregression_amd import kernels as K

dev = "cuda:0"
B, P, F_ = 32, 64, 128
which = sys.argv[1] if len(sys.argv) > 1 else "all"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
x = torch.randn(B, P, P, F_, device=dev).to(torch.bfloat16)
dy = torch.randn(B, P, P, F_, device=dev).to(torch.bfloat16)
w = torch.randn(F_, F_, 3, 3, device=dev) * 0.03
pack = K.pack_conv(w, 0, K.BF16, frag=True)      # (the order the engine hands the heads' weights over in: csrc/conv_wstat.hip)
pack_d = K.pack_conv(w, 1, K.BF16)                # (data gradients: the patch kernel, standard order)
st = K.norm_stats(x, torch.ones(F_, device=dev), torch.zeros(F_, device=dev), mode=0)
bias = torch.zeros(F_, device=dev)
flops = 2.0 * B * P * P * F_ * F_ * 9

def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3

if which in ("all", "fwd"):
    t = timeit(lambda: K.conv_fwd(x, pack, F_, 3, 1, bias=bias, norm=st))
    print(json.dumps({"kernel": "conv3x3_wstat fwd (+NR prologue)", "us": t * 1e6, "TFLOPs": flops / t / 1e12}))
    t = timeit(lambda: K.conv_fwd(x, pack, F_, 3, 1))
    print(json.dumps({"kernel": "conv3x3_wstat (no prologue)", "us": t * 1e6, "TFLOPs": flops / t / 1e12}))
    ys = torch.randn(B, P, P, F_, device=dev).to(torch.bfloat16)
    t = timeit(lambda: K.conv_fwd_stats(x, pack_d, F_, 3, 1, nb_y=ys, nb_state=st))
    print(json.dumps({"kernel": "conv3x3_patch data gradient + norm-backward sums (incl. the NaN fill of the test wrapper)", "us": t * 1e6, "TFLOPs": flops / t / 1e12}))
    wj = torch.randn(14, F_, 3, 3, device=dev) * 0.03
    pack_j = K.pack_conv(wj, 0, K.BF16)
    t = timeit(lambda: K.conv_fwd(x, pack_j, 14, 3, 1, bias=bias[:14].contiguous(), norm=st, nhwc_out=False, nchw_out=True))
    print(json.dumps({"kernel": "conv3x3_wstat narrow form: 128 -> 14, fp32 NCHW out (+NR prologue)", "us": t * 1e6, "TFLOPs": flops * 14 / 128 / t / 1e12}))
if which in ("all", "wgrad"):
    # the engine's split count for this layer (80) only, so that the rocprofv3 average of these launches is the number quoted in
    # DESIGN.md; `sweep` as third argument walks the split counts (round 2's 81 us "average" was over such a sweep)
    sweep = len(sys.argv) > 3 and sys.argv[3] == "sweep"
    # (round 4: these layers run on conv_wgrad3w_kernel, the wave-specialised kernel; 24 splits = the engine's count in the train step -- few,
    # long workgroups that leave the other CUs to the chain -- 80 = the one-round-of-the-chip count of the isolated comparison)
    # third argument: `sweep`, or ONE split count (so that a rocprofv3 average covers one configuration); default 80
    arg3 = sys.argv[3] if len(sys.argv) > 3 else "80"
    for splits in ((24, 40, 57, 80, 96, 120, 160) if sweep else (int(arg3),)):
        t = timeit(lambda: K.conv_wgrad(x, dy, F_, 3, 1, norm=st, splits=splits))
        print(json.dumps({"kernel": "conv_wgrad3w<norm> (wave-specialised, norm + ReLU in LDS) + reduce, splits %d" % splits, "us": t * 1e6, "TFLOPs": flops / t / 1e12}))
        t = timeit(lambda: K.conv_wgrad(x, dy, F_, 3, 1, norm=None, splits=splits))
        print(json.dumps({"kernel": "conv_wgrad3w<no norm> + reduce, splits %d" % splits, "us": t * 1e6, "TFLOPs": flops / t / 1e12}))
