"""FIRST file of the GPU suite: per-kernel known-answer digests (tests/kat_cases.py, tests/golden/kat_digests.json).

Every reduction of the library has a fixed order, so on a healthy MI355X the bytes each kernel writes for fixed input bytes are a
constant of the build.  A box (or a build) that computes anything else is named here, kernel family by kernel family, before the
parity tests run: what a mismatch means and what to do next is DESIGN.md section 2 ("Known-answer digests").
Regenerate after a deliberate change of arithmetic:  python tools/kat.py --write  (on the GPU box).
"""
import json
import os

import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kat_digests.json")


def test_kernel_known_answer_digests():
    import kat_cases
    want = json.load(open(GOLDEN))["digests"]
    got = kat_cases.compute()
    missing = sorted(set(kat_cases.CASES) - set(want))
    assert not missing, "cases without a committed digest (run tools/kat.py --write on a GPU box): %s" % missing
    bad = sorted(n for n in got if got[n] != want[n])
    fams = sorted({n.split("/")[0] for n in bad})
    assert not bad, ("%d of %d known-answer cases differ on this box / build.  Kernel families: %s.  Cases: %s.  "
                     "Next: python tools/kat.py --diagnose (repeats the differing cases, then tools/lease_check.py --bisect and "
                     "tools/race_hunt.py)" % (len(bad), len(got), fams, bad))


def test_a_broken_digest_names_its_kernel():
    """the checker itself: a deliberately wrong expectation is reported under the case's own name"""
    import kat_cases
    name = "pool/maxpool_upsample_colsum"
    got = kat_cases.compute([name])
    assert list(got) == [name] and got[name] != "0" * 64
    again = kat_cases.compute([name])
    assert again == got, "the same case twice on one box must give the same digest"
