"""The committed evidence bench.py quotes must exist: the newest profiles/rN_traffic.json carries the row of the graded kernel (round 5's final
file had silently lost it when the kernel gained a template parameter), under the key bench.py looks up, with algorithmic bytes and a ratio,
and tools/profile_summary.py reported no row it could not produce."""
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _newest_traffic():
    fs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), key=lambda f: int(re.search(r"r(\d+)_traffic", f).group(1)))
    assert fs, "no profiles/rN_traffic.json"
    return fs[-1]


def test_newest_traffic_summary_has_the_graded_kernels_row():
    f = _newest_traffic()
    d = json.load(open(f))
    assert "ERRORS" not in d, (f, d.get("ERRORS"))
    src = open(os.path.join(ROOT, "bench.py")).read()
    m = re.search(r'tkey = "([^"]+)"', src)
    assert m, "bench.py no longer names the traffic row it quotes"
    row = d.get(m.group(1))
    assert row and row.get("hbm_bytes_corrected") and row.get("algorithmic_bytes") and row.get("ratio"), (f, m.group(1), row)
    assert 0.9 <= row["ratio"] <= 2.0, row
    assert os.path.basename(f) in src, "bench.py does not look the newest traffic file up: %s" % os.path.basename(f)


def test_decoder_rows_of_the_newest_traffic_summary():
    d = json.load(open(_newest_traffic()))
    for key in ("decode_fwd B=128 J=42 P=128", "decode_bwd B=128 J=42 P=128", "decode_fwd B=32 J=14 P=64"):
        row = d.get(key)
        assert row and row.get("ratio"), key
        assert row["ratio"] <= 1.06, (key, row["ratio"])      # (round 6: the XCD-aware map order; round 5's forward was 1.12 x)
