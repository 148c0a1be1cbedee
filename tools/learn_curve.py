"""Does the bf16 (throughput) engine LEARN like the fp32 (parity) engine?  Trains BASELINE C2's architecture with the
reference's loop (train.py:158-212: 3-term loss, AdamW, lr 1e-3 unless given) on a STREAM of rendered synthetic hands
(pixelwiseregression_amd.synthetic.make_pose_batch: a new seed per step, targets are a function of the image), once per
precision from the same initial weights, and evaluates held-out batches with the reference's validation metric
(train.py:254-285: mean 3D joint error in mm, NYU intrinsics) every `--eval-every` steps.

    python tools/learn_curve.py --steps 600 --out profiles/r2_learning.json
"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


from pixelwiseregression_amd.evaluate import train_and_validate as run


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--eval-every", type=int, default=100)
    ap.add_argument("--precisions", default="bf16,fp32")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    res = {}
    for prec in a.precisions.split(","):
        res[prec] = run(prec, a.steps, lr=a.lr, eval_every=a.eval_every, log=print)
    if "bf16" in res and "fp32" in res:
        r = res["bf16"]["final_mm"] / res["fp32"]["final_mm"]
        print("final mm error: bf16 %.2f  fp32 %.2f  ratio %.3f" % (res["bf16"]["final_mm"], res["fp32"]["final_mm"], r))
        res["bf16_over_fp32"] = r
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        json.dump(res, open(a.out, "w"))
