"""Interleaved A/B of the train step under switches of the DEBUG build (tools/build_debug.py), on ONE lease: every variant is a fresh child
process (the switches are read once per process), variants alternate `rounds` times, each run = 20 warm-up + `steps` timed steps of the
bench configuration (BASELINE C2, B = 32, bf16, AdamW) followed by an inference loop; the 220-step loss digest of every variant is printed
beside its times (bit-identical kernels give the same value).
    python tools/ab_step.py [--steps 200] [--rounds 2] "NAME=VAL ..." "NAME=VAL ..." ...      ("-" = no switch: the default configuration)
    python tools/ab_step.py --child            (internal: one run, prints one JSON line)"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(steps):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
    import dbglib  # noqa: F401
    import torch
    from pixelwiseregression_amd import PixelwiseRegression
    from pixelwiseregression_amd.synthetic import make_batch
    from pixelwiseregression_amd.train import TrainStep
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    m = PixelwiseRegression(14, stage=2, label_size=64, features=128, level=4, norm_method="instance").to(dev).set_precision("bf16").train()
    ts = TrainStep(m, opt="adam", lr=1e-4)
    b = make_batch(32, 14, S=128, seed=1234, device=dev)
    a = (b["img"], b["label_img"], b["mask"], b["uvd"])
    for _ in range(20):
        loss = ts(*a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = ts(*a)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    final = float(loss.item())
    m.eval()
    with torch.no_grad():
        for _ in range(5):
            m(a[0], a[1], a[2])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            m(a[0], a[1], a[2])
        torch.cuda.synchronize()
        inf = (time.perf_counter() - t0) / 100 * 1e3
    print(json.dumps({"ms_per_step": round(ms, 4), "infer_ms": round(inf, 4), "loss_after_%d" % (20 + steps): final}))


if __name__ == "__main__":
    args = sys.argv[1:]
    steps, rounds = 200, 2
    if "--steps" in args:
        i = args.index("--steps"); steps = int(args[i + 1]); del args[i:i + 2]
    if "--rounds" in args:
        i = args.index("--rounds"); rounds = int(args[i + 1]); del args[i:i + 2]
    if "--child" in args:
        child(steps)
        sys.exit(0)
    variants = args or ["-"]
    res = {v: [] for v in variants}
    for r in range(rounds):
        for v in variants:
            env = dict(os.environ)
            for kv in ([] if v == "-" else v.split()):
                k, val = kv.split("=", 1)
                env[k] = val
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--steps", str(steps)], env=env, capture_output=True, text=True)
            line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
            res[v].append(json.loads(line[-1]) if line else {"error": (p.stderr or p.stdout)[-400:]})
            print(json.dumps({"variant": v, "round": r, **res[v][-1]}), flush=True)
    print(json.dumps({"summary": {v: {"ms_per_step": [x.get("ms_per_step") for x in rs], "infer_ms": [x.get("infer_ms") for x in rs]} for v, rs in res.items()}}))
