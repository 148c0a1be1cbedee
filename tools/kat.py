"""Known-answer digests of every kernel family (tests/kat_cases.py).
   python tools/kat.py            compare with tests/golden/kat_digests.json, print the names that differ (exit 1 on a mismatch)
   python tools/kat.py --write    write tests/golden/kat_digests.json from this box (after a deliberate change of arithmetic)
   python tools/kat.py --diagnose on a mismatch: each differing case five more times (a box that disagrees with ITSELF is a race or a
                                  marginal part; one that repeats its own wrong answer computes differently), then tools/lease_check.py
                                  --bisect and tools/race_hunt.py"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
GOLDEN = os.path.join(ROOT, "tests", "golden", "kat_digests.json")

if __name__ == "__main__":
    import torch
    import kat_cases
    got = kat_cases.compute()
    if "--write" in sys.argv:
        rec = {"device": torch.cuda.get_device_name(0), "torch": torch.__version__, "digests": got}
        json.dump(rec, open(GOLDEN, "w"), indent=1, sort_keys=True)
        print("wrote %d digests to %s" % (len(got), GOLDEN))
        sys.exit(0)
    want = json.load(open(GOLDEN))["digests"]
    bad = sorted(n for n in got if want.get(n) != got[n])
    print(json.dumps({"cases": len(got), "differ": bad}))
    if bad and "--diagnose" in sys.argv:
        for n in bad:
            reps = [kat_cases.compute([n])[n] for _ in range(5)]
            print("%s: first %s, five repeats %s" % (n, got[n][:12], "all equal to it" if all(r == got[n] for r in reps) else [r[:12] for r in reps]), flush=True)
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "lease_check.py"), "--bisect"])
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "race_hunt.py")])
    sys.exit(1 if bad else 0)
