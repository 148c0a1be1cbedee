"""Step-level timing by ELIMINATION (round-3 review item 1a): the train step of BASELINE configs[1] (B=32, bf16, AdamW) with whole
families of launches removed, to fix the ceiling of each lever.  Runs on the DEBUG build (tools/build_debug.py); with launches skipped
the RESULTS ARE WRONG BY CONSTRUCTION -- only the times mean anything.  One child process per variant (the switches are read once).

    python tools/step_elimination.py [steps] > profiles/r3_step_elimination.json

variants: PWR_ELIM bit 0 = no parameter-gradient launches (everything the side streams do), bit 1 = no norm-backward launches,
bit 2 = no data-gradient convs / fused ResBlock backwards; PWR_SIDE_STREAM=0 = everything on the caller's stream (the serial sum);
PWR_SIDE_CUS=n = side streams confined to n CUs by a CU mask (and, `chain_cus`, the chain on a masked stream of its own).
"""
import json, os, subprocess, sys, time

HERE = os.path.dirname(os.path.abspath(__file__))


def child():
    import ctypes
    import torch
    sys.path.insert(0, HERE)
    import dbglib  # noqa: F401
    from pixelwiseregression_amd import PixelwiseRegression
    from pixelwiseregression_amd.synthetic import make_batch
    from pixelwiseregression_amd.train import TrainStep
    steps = int(os.environ.get("ELIM_STEPS", "100"))
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    m = PixelwiseRegression(14, stage=2, label_size=64, features=128, level=4, norm_method="instance").to(dev).set_precision("bf16").train()
    tr = TrainStep(m, opt="adam", lr=1e-4, beta1=0.9, beta2=0.999, weight_decay=0.0, alpha=1.0, lambda_h=1.0, lambda_d=0.01)
    b = make_batch(32, 14, S=128, seed=1234, device=dev, dense_targets=True)
    step = lambda: tr(b["img"], b["label_img"], b["mask"], b["uvd"], b["heatmaps"], b["depthmaps"])
    ctx = None
    chain_cus = int(os.environ.get("ELIM_CHAIN_CUS", "0"))
    if chain_cus:
        # the chain on a CU-masked stream of its own: mask = the LAST chain_cus bits (the side streams take the first ones)
        hip = ctypes.CDLL("libamdhip64.so")
        mask = (ctypes.c_uint32 * 8)()
        for i in range(chain_cus):
            bit = 255 - i
            mask[bit >> 5] |= 1 << (bit & 31)
        sp = ctypes.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(sp), 8, mask)
        assert rc == 0, rc
        ctx = torch.cuda.stream(torch.cuda.ExternalStream(sp.value, device=dev))
        ctx.__enter__()
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    # forward-only (training plan) and whole step, each timed over `steps`
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    if ctx is not None:
        ctx.__exit__(None, None, None)
    print("ELIM_RESULT " + json.dumps({"ms_per_step": dt * 1e3}))


VARIANTS = [
    ("baseline (debug build, no switch)", {}),
    ("no weight-gradient / parameter-gradient launches (side streams idle)", {"PWR_ELIM": "1"}),
    ("no norm-backward launches (norm_bwd_sum + norm_bwd_apply)", {"PWR_ELIM": "2"}),
    ("neither", {"PWR_ELIM": "3"}),
    ("no data-gradient convs / fused ResBlock backwards", {"PWR_ELIM": "4"}),
    ("forward + loss + AdamW only (no backward kernels at all)", {"PWR_ELIM": "7"}),
    ("no fork events: side launches NOT ordered behind the chain (what the event records cost the chain)", {"PWR_ELIM": "8"}),
    ("no chain backward kernels, side work only (forward + parameter gradients)", {"PWR_ELIM": "6"}),
    ("everything on the caller's stream (no side streams): the serial sum", {"PWR_SIDE_STREAM": "0"}),
    ("one side stream", {"PWR_SIDE_STREAM": "1"}),
    ("side streams on 64 CUs (first 64 mask bits), chain unmasked", {"PWR_SIDE_CUS": "64"}),
    ("side streams on 64 CUs (every 4th mask bit), chain unmasked", {"PWR_SIDE_CUS": "64", "PWR_SIDE_CU_PATTERN": "1"}),
    ("side streams on 128 CUs (every 2nd mask bit), chain unmasked", {"PWR_SIDE_CUS": "128", "PWR_SIDE_CU_PATTERN": "1"}),
    ("side 64 CUs (first bits) / chain 192 CUs (last bits)", {"PWR_SIDE_CUS": "64", "ELIM_CHAIN_CUS": "192"}),
    ("side 128 CUs / chain 128 CUs", {"PWR_SIDE_CUS": "128", "ELIM_CHAIN_CUS": "128"}),
    ("side 192 CUs / chain 64 CUs", {"PWR_SIDE_CUS": "192", "ELIM_CHAIN_CUS": "64"}),
    ("side unmasked / chain on a 256-CU masked stream (cost of the masked-stream path itself)", {"ELIM_CHAIN_CUS": "256"}),
    ("baseline again (drift of the box)", {}),
]

if __name__ == "__main__":
    if os.environ.get("ELIM_CHILD"):
        child()
        sys.exit(0)
    steps = sys.argv[1] if len(sys.argv) > 1 else "100"
    out = {"what": "train step, BASELINE configs[1] (B=32, bf16, AdamW), debug build, %s timed steps per variant after 20 warm-up; "
                   "results of the elimination variants are wrong by construction" % steps, "variants": []}
    only = os.environ.get("ELIM_ONLY")          # e.g. "0,6,7": a subset of the variants by index
    sel = [int(i) for i in only.split(",")] if only else range(len(VARIANTS))
    for name, env in [VARIANTS[i] for i in sel]:
        e = dict(os.environ, ELIM_CHILD="1", ELIM_STEPS=steps, **env)
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=e, capture_output=True, text=True, timeout=600)
        ms = None
        for line in r.stdout.splitlines():
            if line.startswith("ELIM_RESULT "):
                ms = json.loads(line[len("ELIM_RESULT "):])["ms_per_step"]
        rec = {"variant": name, "env": env, "ms_per_step": ms}
        if ms is None:
            rec["error"] = (r.stderr or r.stdout)[-400:]
        out["variants"].append(rec)
        print(json.dumps(rec), file=sys.stderr, flush=True)
    print(json.dumps(out, indent=1))
