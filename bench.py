#!/usr/bin/env python3
"""bench.py -- depth-frames/sec of the PixelwiseRegression hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one TRAINING step of BASELINE.json configs[1] (NYU shape: 14 joints, 128x128 crops, batch 32 per GPU,
features 128, level 4, stage 2, instance norm, bf16 engine): forward + the 3-term loss of train.py:195-205
+ backward + AdamW(weight_decay=0) step (train.py:139-140), on seeded synthetic crops that are already
resident in HBM.  Weak scaling: every rank processes its own 32-frame shard; gradients are all-reduced
over RCCL per backward segment.  Rank 0 prints ONE JSON line; besides the contract fields it carries
  roofline      -- the dominant kernel (the 128->128 3x3 implicit-GEMM conv of the heads: 72 % of the conv
                   FLOPs), timed live with HIP events at the workload's shape, against the dense bf16 MFMA peak
  roofline_decoder -- the soft-argmax decoder (HBM bound) forward at the same batch, for reference
  cpu_baseline  -- the CPU oracle ("port" of the reference's op sequence) timed on this host's cores, BASELINE.md section 3
                   protocol (config batch, os.cpu_count() threads, 3 warm-up + 10 timed steps, median)
  infer_frames_per_s -- forward+decode only (no_grad), same batch
  accuracy      -- the second half of BASELINE.json's metric: mean 3D joint error (mm, train.py:254-285) on held-out rendered
                   synthetic hands after --accuracy-steps training steps, for the bf16 engine (the benched configuration) and
                   the fp32 parity engine from the same initial weights (outside the timed region)
  step_mfma_frac -- conv FLOPs of one train step (3 x 20.88 GFLOP/frame) / step time / dense bf16 MFMA peak
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

J, S, P, F_, LEVEL, STAGE, B_PER_GPU = 14, 128, 64, 128, 4, 2, 32
PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA peak, MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3        # f32-input MFMA (= the fp32 vector peak), same guide
PEAK_HBM_GBS = 8000.0


def train_loss(results, batch, alpha=1.0, lambda_h=1.0, lambda_d=0.01):
    loss = 0
    for (heat, depth, uvd) in results:                      # train.py:195-205
        hl = lambda_h * torch.mean(torch.sum((heat - batch["heatmaps"]) ** 2, dim=(2, 3)))
        dl = lambda_d * torch.mean(torch.sum((depth - batch["depthmaps"]) ** 2, dim=(2, 3)))
        ul = torch.mean(torch.sum((uvd - batch["uvd"]) ** 2, dim=2))
        loss = loss + alpha * ul + (1 - alpha) * (hl + dl)
    return loss


def time_head_conv(dev, B, iters=20, precision="bf16"):
    """The dominant kernel at the workload's shape: x [B,64,64,128] -> conv3x3 128->128 (+ fused norm/ReLU prologue), in the
    engine's precision (bf16 MFMA, or the exact-f32 MFMA of the parity mode)."""
    from pixelwiseregression_amd import kernels as K
    x = torch.randn(B, P, P, F_, device=dev)
    if precision == "bf16":
        x = x.to(torch.bfloat16)
    w = torch.randn(F_, F_, 3, 3, device=dev) * 0.03
    # (bf16: the pack in the order the engine hands this layer's weights to the kernel -- csrc/conv_wstat.hip's fragment order)
    pack = K.pack_conv(w, 0, K.BF16, frag=True) if precision == "bf16" else K.pack_conv(w, 0, K.F32)
    gamma, beta = torch.ones(F_, device=dev), torch.zeros(F_, device=dev)
    st = K.norm_stats(x, gamma, beta, mode=0)
    bias = torch.zeros(F_, device=dev)
    for _ in range(3):
        K.conv_fwd(x, pack, F_, 3, 1, bias=bias, norm=st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        K.conv_fwd(x, pack, F_, 3, 1, bias=bias, norm=st)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / iters * 1e-3
    flops = 2.0 * B * P * P * F_ * F_ * 9
    return t, flops


def vendor_gemm_yardstick(dev, B, iters=30):
    """Calibration, not a product path: the vendor GEMM (hipBLASLt through torch.matmul, bf16) on this GPU at the dominant conv's
    shape seen as an implicit GEMM -- M = B*64*64 pixels, N = 128 output channels, K = 9*128 -- i.e. what a tuned library kernel
    reaches at N = 128 WITHOUT the im2col gather, the norm / ReLU prologue and the statistics epilogue the conv kernel carries."""
    M, N, K = B * P * P, F_, 9 * F_
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    b = torch.randn(K, N, device=dev, dtype=torch.bfloat16)
    for _ in range(5):
        a @ b
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        a @ b
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / iters * 1e-3
    return {"what": "torch.matmul bf16 [%d x %d] @ [%d x %d] (hipBLASLt), same FLOPs as the conv launch" % (M, K, K, N),
            "us": t * 1e6, "TFLOP/s": 2.0 * M * N * K / t / 1e12, "frac": 2.0 * M * N * K / t / 1e12 / PEAK_BF16_TFLOPS}


def time_decoder(dev, B, iters=50):
    """The decoder forward at the workload's shape, launched back to back through the C ABI on preallocated buffers (going through
    ops.decode_forward -- two torch.empty per call -- made the measurement host-bound: 11.8 us per call for a 7.6 us kernel)."""
    from pixelwiseregression_amd import _lib
    l = _lib.lib()
    z = torch.randn(B, J, P, P, device=dev)
    D = torch.randn(B, J, P, P, device=dev)
    m = (torch.rand(B, 1, P, P, device=dev) < 0.4).float()
    L = torch.randn(B, 1, P, P, device=dev) * m
    w = torch.ones(J, device=dev)
    p_out, uvd = torch.empty_like(z), torch.empty(B, J, 3, device=dev)
    s = _lib.stream_ptr(dev)

    def launch():
        _lib.check(l.pwr_decode_fwd(z.data_ptr(), D.data_ptr(), L.data_ptr(), m.data_ptr(), w.data_ptr(), p_out.data_ptr(), uvd.data_ptr(),
                                    B, J, P, 0, s), "pwr_decode_fwd")
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        launch()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / iters * 1e-3
    nbytes = 12.0 * B * J * P * P + 8.0 * B * P * P + 12.0 * B * J
    return t, nbytes


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _usable_cpus():
    """CPUs this process may really use: the affinity mask, capped by the cgroup CPU quota (os.cpu_count() reports the host's
    256 logical CPUs inside a container that is allowed far fewer; 256 torch threads then run 100x slower than 32)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        if q[0] != "max":
            n = min(n, max(1, int(float(q[0]) / float(q[1]) + 0.5)))
    except (OSError, ValueError, IndexError):
        try:
            q, per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()), int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / per + 0.5)))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(Bc=B_PER_GPU, warmup=3, iters=10, budget_s=60.0):
    """CPU oracle on a bounded sample of the same workload, BASELINE.md section 3 protocol: the config's batch, fp32,
    torch.set_num_threads(os.cpu_count()), `warmup` + `iters` timed train steps (fwd + loss of train.py:197-205 + bwd + AdamW),
    median; inference (no_grad forward + decode) separately.  If the host is so slow that this would exceed `budget_s`, the
    iteration counts are cut (never below 1 + 3) and the `sample` text says so."""
    import statistics
    from oracle import model_ref
    from pixelwiseregression_amd import PixelwiseRegression
    from pixelwiseregression_amd.synthetic import make_batch
    torch.manual_seed(0)
    m = PixelwiseRegression(J, stage=STAGE, label_size=P, features=F_, level=LEVEL, norm_method="instance")
    params = {k: v.detach().clone().requires_grad_(v.is_floating_point() and "filter" not in k) for k, v in m.state_dict().items()}
    opt = torch.optim.AdamW([v for v in params.values() if v.requires_grad], lr=1e-4, weight_decay=0)
    rc = model_ref.RefConfig(J, STAGE, P, F_, LEVEL, 3, "instance", "softmax")
    batch = make_batch(Bc, J, S=S, seed=1234, dense_targets=True)

    def step():
        t = time.perf_counter()
        opt.zero_grad()
        res = model_ref.forward(params, rc, batch["img"], batch["label_img"], batch["mask"], training=True)
        loss = model_ref.train_loss(res, batch["uvd"], batch["heatmaps"], batch["depthmaps"], alpha=1.0)
        loss.backward()
        opt.step()
        return time.perf_counter() - t

    def infer():
        t = time.perf_counter()
        with torch.no_grad():
            model_ref.forward(params, rc, batch["img"], batch["label_img"], batch["mask"], training=False)
        return time.perf_counter() - t

    # Threads: BASELINE.md says os.cpu_count(); on a many-core host (or a container with a CPU quota) that many threads are far
    # SLOWER than fewer for this batch size, so the thread count is picked by a short scan over the usable CPUs (one inference
    # forward each, the fastest wins) and reported as `cores` = the threads actually used.
    usable = _usable_cpus()
    cands = sorted({c for c in (8, 16, 32, 64, 96, 128, usable) if c <= usable} or {usable})
    scan = {}
    for c in cands:
        torch.set_num_threads(c)
        infer()
        scan[c] = infer()
        if c > cands[0] and scan[c] > 1.5 * min(scan.values()):
            break                                  # past the knee: more threads only get slower
    cores = min(scan, key=scan.get)
    torch.set_num_threads(cores)
    t_first = step()
    if t_first * (warmup + iters) > budget_s:
        warmup, iters = 1, max(3, min(iters, int(budget_s / t_first) - 1))
    for _ in range(warmup - 1):
        step()
    ts = [step() for _ in range(iters)]
    dt = statistics.median(ts)
    infer()
    ti = statistics.median([infer() for _ in range(3)])
    return {"value": Bc / dt, "unit": "frames/s", "cores": cores, "kind": "port", "cpu": _cpu_model(),
            "os_cpu_count": os.cpu_count(), "usable_cpus": usable, "thread_scan_s_per_forward": {str(k): round(v, 3) for k, v in scan.items()},
            "infer_value": Bc / ti,
            "sample": "median of %d timed train steps (fwd+loss+bwd+AdamW) after %d warm-up of the CPU oracle at batch %d (the config's), "
                      "same architecture and 128x128 crops, fp32, %d threads (fastest of a scan over the usable CPUs); inference = median of 3 no_grad forwards. "
                      "Oracle vs the reference itself (build container, 8 threads, B=8, interleaved rounds, profiles/r4_ref_vs_oracle_cpu.json): the same ATen "
                      "ops in the same counts (torch.profiler), inference 0.98x, train 0.91x of the reference's speed (per round 0.92-1.05 / 0.89-1.01: noise of "
                      "a shared container; round 3's 0.79x was a single reference-first, oracle-second pass)"
                      % (iters, warmup, Bc, cores)}


def accuracy_block(steps, dev):
    """mean 3D joint error (mm) after `steps` training steps on streamed synthetic hands, bf16 vs fp32 engine (same init)."""
    from pixelwiseregression_amd.evaluate import train_and_validate
    out = {"steps": steps, "task": "rendered synthetic hands (synthetic.make_pose_batch), NYU intrinsics, cube 150 mm, batch %d, AdamW lr 1e-3, "
                                   "4 held-out batches; metric = train.py:254-285" % B_PER_GPU}
    for prec in ("bf16", "fp32"):
        r = train_and_validate(prec, steps, B=B_PER_GPU, J=J, S=S, lr=1e-3, eval_every=max(50, steps // 4), dev=dev, features=F_, level=LEVEL,
                               stage=STAGE)
        out["mm_error_" + prec] = r["final_mm"]
        out["mm_error_untrained"] = r["curve"][0]["mm"][-1]
        out["curve_" + prec] = [(c["step"], round(c["mm"][-1], 3)) for c in r["curve"]]
        out["final_train_loss_" + prec] = sum(r["train_loss"][-20:]) / 20
    out["bf16_over_fp32"] = out["mm_error_bf16"] / out["mm_error_fp32"]
    return out


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as FRESH child processes (torch.distributed.run), before this
    process has touched the GPU, relay rank 0's JSON line and return non-zero if any rank failed.  (Never a re-exec of a process that
    has initialised HIP.)"""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n, "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this host driver
    # Host side of a rank: ONE Python thread issues the whole step (6 ms of ctypes calls per 6 ms step: `rccl.host_issue_ms_per_step`), plus
    # RCCL's proxy thread -- two busy cores per rank, no OpenMP work at all.  Two OpenMP threads leave torch's few CPU ops room without
    # oversubscribing a 16-CPU box at 8 ranks.
    env.setdefault("OMP_NUM_THREADS", "2")
    # RCCL channel budget: the three all-reduces of a step move 5.8 + 5.8 + 0.9 MiB (latency-bound on xGMI); every channel is a persistent
    # workgroup that takes a CU from the full-chip convs it runs beside.  4 channels are plenty for 6 MiB and cost at most 4 of 256 CUs.
    env.setdefault("NCCL_MAX_NCHANNELS", "4")
    env.setdefault("NCCL_MIN_NCHANNELS", "1")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for l_ in proc.stdout:
        if l_.startswith("{") and '"metric"' in l_:
            line = l_.rstrip("\n")
        else:
            sys.stderr.write(l_)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    if rc == 0 and line is None:
        print("bench.py: the ranks exited cleanly but rank 0 printed no result line", file=sys.stderr)
        rc = 1
    return rc


def pipeline_block(model, trainer, dev, steps):
    """End to end: RAW 640x480 depth frames (resident in HBM) -> preprocess_batch (crop around the COM, depth cut, resize, the rotation /
    scale / shift augmentation drawn per step like datasets.py:224-238, label image, mask, normalisation; csrc/preprocess.hip) ->
    TrainStep, every step.  Reported beside `value`, never as it (the bench's step starts from crops already in HBM)."""
    import random
    from pixelwiseregression_amd import preprocess_batch, draw_augmentation, INTRINSICS
    from pixelwiseregression_amd.synthetic import make_raw_frames
    raw = make_raw_frames(B_PER_GPU, J, seed=4321, device=dev)
    rng = random.Random(99)

    def prep():
        return preprocess_batch(raw["depth"], raw["joint_uvd"], raw["com"], raw["cube_size"], INTRINSICS["NYU"], S, P,
                                augmentation=draw_augmentation(B_PER_GPU, rng=rng), dense_targets=False)

    def both():
        b = prep()
        return trainer(b["img"], b["label_img"], b["mask"], b["uvd"])
    out = {}
    for name, fn in (("preprocess_only", prep), ("preprocess_plus_train_step", both)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            r = fn()
        torch.cuda.synchronize()
        out[name + "_ms"] = (time.perf_counter() - t0) / steps * 1e3
    out["frames_per_s"] = B_PER_GPU / (out["preprocess_plus_train_step_ms"] * 1e-3)
    last = prep()
    out["fallback_or_rejected_in_one_batch"] = int((last["fallback"] | last["rejected"].cpu()).sum())      # (one fresh draw: informational)
    out["what"] = ("%d raw 480x640 fp32 frames in HBM -> preprocess_batch with a fresh augmentation draw per step -> TrainStep; %d timed steps; the "
                   "per-sample crop geometry and joint transforms are float64 host arithmetic like the reference's" % (B_PER_GPU, steps))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--roofline-only", action="store_true",
                    help="run only the two roofline probes (dominant conv, decoder) and print their JSON: the command profiles/ holds a "
                         "rocprofv3 --kernel-trace --stats summary of, so that the live HIP-event time can be checked against rocprofv3's average")
    ap.add_argument("--accuracy-steps", type=int, default=300,
                    help="training steps of the accuracy block (mm error, bf16 and fp32 engines; rank 0 at N=1 only); 0 = skip")
    ap.add_argument("--harness", default="native", choices=["native", "torch"],
                    help="native: pixelwiseregression_amd.train.TrainStep (loss + AdamW kernels on the flat buffers); "
                         "torch: autograd + torch.optim.AdamW, the reference's loop verbatim")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, the real path) | gloo (debug)")
    ap.add_argument("--force-dist", action="store_true",
                    help="debug: take the data-parallel path (process group, per-segment all-reduce) even with one rank -- exercises RCCL on a 1-GPU box")
    ap.add_argument("--same-device", action="store_true",
                    help="debug: all ranks share cuda:0 (with --dist-backend gloo) to exercise the data-parallel path on a 1-GPU box")
    ap.add_argument("--with-pipeline", action="store_true",
                    help="also time raw frames -> preprocess_batch -> train step (rank 0, N=1; reported as `pipeline`, outside `value`)")
    ap.add_argument("--debug-lib", action="store_true",
                    help="A/B measurements only: load the DEBUG build of the library (tools/build_debug.py) so that the PWR_* experiment "
                         "switches apply; the result line says so")
    args = ap.parse_args()
    if args.debug_lib:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import dbglib  # noqa: F401

    if args.roofline_only:
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        peak = PEAK_BF16_TFLOPS if args.precision == "bf16" else PEAK_F32_TFLOPS
        t, flops = time_head_conv(dev, B_PER_GPU, precision=args.precision)
        td, nb = time_decoder(dev, B_PER_GPU)
        print(json.dumps({"roofline": {"bound": "mfma", "achieved": flops / t / 1e12, "peak": peak, "unit": "TFLOP/s", "frac": flops / t / 1e12 / peak,
                                       "us_per_launch": t * 1e6, "launches": 23},
                          "roofline_decoder": {"bound": "hbm", "achieved": nb / td / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                               "frac": nb / td / 1e9 / PEAK_HBM_GBS, "us_per_launch": td * 1e6, "launches": 53}}), flush=True)
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: become one.  The ranks are FRESH child processes (no exec of this one), so it does not matter whether
        # device_count() initialised HIP here (it may: without amdsmi it falls back to hipGetDeviceCount).
        if not args.same_device and torch.cuda.device_count() < args.gpus:
            print("bench.py --gpus %d: only %d GPU(s) visible (add --same-device --dist-backend gloo for the 1-GPU debug mode)"
                  % (args.gpus, torch.cuda.device_count()), file=sys.stderr)
            sys.exit(2)
        sys.exit(self_launch(args.gpus))
    if args.gpus != world and not args.force_dist:
        print("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    if args.same_device:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        # (also when the driver's torch.distributed.run started this rank: the defaults of self_launch(), read by RCCL at communicator creation)
        os.environ.setdefault("NCCL_MAX_NCHANNELS", "4")
        os.environ.setdefault("NCCL_MIN_NCHANNELS", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)

    from pixelwiseregression_amd import PixelwiseRegression
    from pixelwiseregression_amd.synthetic import make_batch
    torch.manual_seed(0)
    model = PixelwiseRegression(J, stage=STAGE, label_size=P, features=F_, level=LEVEL, norm_method="instance")
    model = model.to(dev).set_precision(args.precision).train()
    if use_dist:
        from pixelwiseregression_amd.ddp import DataParallel
        DataParallel(model)
    native = args.harness == "native"
    if native:
        from pixelwiseregression_amd.train import TrainStep
        trainer = TrainStep(model, opt="adam", lr=1e-4, beta1=0.9, beta2=0.999, weight_decay=0.0, alpha=1.0, lambda_h=1.0, lambda_d=0.01)
    else:
        flat = torch.nn.Parameter(model.flat_parameters())
        flat.grad = model.flat_grad()
        opt = torch.optim.AdamW([flat], lr=1e-4, betas=(0.9, 0.999), weight_decay=0, fused=True)
    batch = make_batch(B_PER_GPU, J, S=S, seed=1234 + rank, device=dev, dense_targets=True)

    def step():
        if native:
            return trainer(batch["img"], batch["label_img"], batch["mask"], batch["uvd"], batch["heatmaps"], batch["depthmaps"])
        model.zero_grad(set_to_none=True)
        res = model(batch["img"], batch["label_img"], batch["mask"])
        loss = train_loss(res, batch)
        loss.backward()
        opt.step()
        return loss

    def barrier():
        if use_dist:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    # host time to ISSUE one step (the caller's thread returns before the GPU has finished): what one rank's Python thread needs per step
    host_issue_ms = None
    if args.warmup > 0:
        th = time.perf_counter()
        for _ in range(min(10, args.warmup)):
            step()
        host_issue_ms = (time.perf_counter() - th) / min(10, args.warmup) * 1e3
        barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    final_loss = float(loss.item())

    # inference (forward + decode), same batch
    model.eval()
    with torch.no_grad():
        for _ in range(3):
            model(batch["img"], batch["label_img"], batch["mask"])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n_inf = max(5, args.steps)
        for _ in range(n_inf):
            model(batch["img"], batch["label_img"], batch["mask"])
        torch.cuda.synchronize()
        dt_inf = (time.perf_counter() - t1) / n_inf
        # the same loop with two batches in flight on two streams (serving.py: two plans of the same weights); reported BESIDE the figure above
        dt_inf2 = None
        if world == 1 and native:
            try:      # (an auxiliary figure: whatever goes wrong here must not cost the run its headline line)
                from pixelwiseregression_amd.serving import StreamedInference
                srv = StreamedInference(model, streams=2)
                feed = lambda n: ((batch["img"], batch["label_img"], batch["mask"]) for _ in range(n))
                for _ in srv.run(feed(6)):
                    pass
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in srv.run(feed(2 * n_inf)):
                    pass
                torch.cuda.synchronize()
                dt_inf2 = (time.perf_counter() - t1) / (2 * n_inf)
                del srv
            except Exception as e:      # noqa: BLE001
                print("bench.py: two-stream inference figure skipped: %r" % (e,), file=sys.stderr)
                dt_inf2 = None

    if rank == 0:
        out = {
            "metric": "depth-frames/sec (train step: fwd+loss+bwd+AdamW), NYU 14J 128x128",
            "value": world * B_PER_GPU * args.steps / dt,
            "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: NYU 14-joint, 128x128 depth crops, batch 32 per GPU, train (AdamW, %s), "
                                   "features 128, level 4, stage 2, instance norm" % args.precision,
                       "global_batch": world * B_PER_GPU, "backend": "hip" if not args.debug_lib else "hip (DEBUG build of the library)",
                       "harness": args.harness,
                       "parallelism": "dp%d" % world if use_dist else "single"},
            "infer_frames_per_s": world * B_PER_GPU / dt_inf,
            "infer_frames_per_s_two_streams": (B_PER_GPU / dt_inf2) if dt_inf2 else None,
            "final_loss": final_loss,
        }
        if use_dist:
            # what went over RCCL: the ranks of the process group and the bytes of each per-segment all-reduce (flat fp32 gradient slices in
            # backward-completion order: last stage, ..., stage 0, stem; ddp.py), the host threads each rank was given
            out["rccl"] = {"ranks": torch.distributed.get_world_size(), "backend": args.dist_backend,
                           "allreduce_bytes_per_segment": [4 * (e - b) for (b, e) in model.segment_ranges()],
                           "omp_num_threads": os.environ.get("OMP_NUM_THREADS"), "nccl_max_nchannels": os.environ.get("NCCL_MAX_NCHANNELS"),
                           "host_issue_ms_per_step": host_issue_ms}
        peak = PEAK_BF16_TFLOPS if args.precision == "bf16" else PEAK_F32_TFLOPS
        step_flops = 3 * 20.88e9 * B_PER_GPU            # 20.88 GFLOP/frame forward at C2, x3 for training
        out["step_mfma_frac"] = step_flops / (dt / args.steps) / (peak * 1e12)
        if True:   # (kept as a block: the roofline probes run on every rank-0 report)
            t, flops = time_head_conv(dev, B_PER_GPU, precision=args.precision)
            # HBM bytes per launch are NOT measured in this run (PMC collection needs rocprofv3 passes of its own: tools/profile_r6.sh), so
            # `traffic` is null here; the counter result of the committed profile of the same kernel and shape is quoted beside it
            traffic_profile = None
            tkey = "conv3x3_wstat_kernel<norm prologue, no statistics> B=32 64x64 128->128"
            if args.precision == "bf16":
                # the newest committed counter summary that HAS the row (tools/profile_summary.py fails loudly when a pass loses it: round 5's
                # final file had dropped it after the kernel gained a template parameter)
                for tj in ("r6_traffic.json", "r5_traffic.json"):
                    fp = os.path.join(ROOT, "profiles", tj)
                    v = json.load(open(fp)).get(tkey, {}) if os.path.exists(fp) else {}
                    if v.get("hbm_bytes_corrected") is not None:
                        traffic_profile = {"hbm_bytes_per_launch": v["hbm_bytes_corrected"], "algorithmic_bytes": v.get("algorithmic_bytes"),
                                           "ratio": v.get("ratio"), "source": "profiles/" + tj, "kernel": v.get("kernel", tkey),
                                           "note": "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes on another lease"}
                        break
            out["roofline"] = {"bound": "mfma", "kernel": "%s: conv3x3 128->128 @64x64, B=%d (+fused norm/ReLU), %s operands"
                                                          % ("conv3x3_wstat_kernel<true, 0, 128>" if args.precision == "bf16" else "conv_fwd_kernel<float>", B_PER_GPU, args.precision),
                               "achieved": flops / t / 1e12, "peak": peak, "unit": "TFLOP/s",
                               "frac": flops / t / 1e12 / peak, "traffic": None, "traffic_profile": traffic_profile, "us_per_launch": t * 1e6,
                               "flop_per_launch": flops}
            if args.precision == "bf16":
                out["roofline"]["vendor_gemm_same_shape"] = vendor_gemm_yardstick(dev, B_PER_GPU)
            td, nb = time_decoder(dev, B_PER_GPU)
            dprofile = None
            for tj in ("r6_traffic.json", "r5_traffic.json", "r4_traffic.json"):
                fp = os.path.join(ROOT, "profiles", tj)
                if os.path.exists(fp) and dprofile is None:
                    v = json.load(open(fp)).get("decode_fwd B=%d J=%d P=%d" % (B_PER_GPU, J, P), {}).get("hbm_bytes_corrected")
                    if v is not None:
                        dprofile = {"hbm_bytes_per_launch": v, "source": "profiles/" + tj}
            out["roofline_decoder"] = {"bound": "hbm", "achieved": nb / td / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                       "frac": nb / td / 1e9 / PEAK_HBM_GBS, "traffic": None, "traffic_profile": dprofile,
                                       "us_per_launch": td * 1e6,
                                       "note": "23 MB per launch: launch / latency bound at this shape (the rocprofv3 average of the same probe is ~1 us longer than this "
                                               "HIP-event figure: profiles/r6_rocprofv3_roof_kernel_stats.csv); HBM bound at the C5 shape (profiles/r6_dec_bench.jsonl)"}
        if world == 1 and args.with_pipeline and native:
            model.train()
            out["pipeline"] = pipeline_block(model, trainer, dev, max(10, min(args.steps, 100)))
        if world == 1 and not use_dist and args.accuracy_steps > 0:
            out["accuracy"] = accuracy_block(args.accuracy_steps, dev)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if use_dist:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
