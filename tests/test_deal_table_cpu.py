"""The compile-time schedule of csrc/conv_wstat.hip (wst::make_deal: which staging / epilogue items run behind which MFMA) checked on the
host: the `namespace wst` block is plain constexpr C++, so it is cut out of the kernel source, compiled with g++ and asked for every
(norm prologue, kind) table the kernel instantiates.  A schedule that drops an item, runs one twice or out of order, lets an epilogue item
leave its half tile, or issues a load behind its first use would still compile into a kernel -- one that computes garbage only on the GPU."""
import os
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "pixelwiseregression_amd", "csrc", "conv_wstat.hip")

MAIN = r"""
using namespace wst;
int main() {
  int bad = 0;
  for (int ci = 64; ci <= 128; ci += 64) for (int nrm = 0; nrm < 2; ++nrm) for (int kind = 0; kind < 4; ++kind) {
    if ((kind == 2 && nrm) || (ci == 64 && kind >= 2)) continue;
    const Deal d = make_deal(nrm, kind, ci);
    const bool nar = kind == 3;
    const int ITERS = 9 * (ci / 32), SLOTS = ITERS * 8, HSLOTS = ITERS * 4, NSLOTS = ITERS * 2, NITP = ci == 128 ? 13 : 7;
    const int nsl = nar ? NSLOTS : SLOTS, sv = sv_items(nrm), ns = NITP * sv, eha = e_half(kind, 0, ci), ehb = e_half(kind, 1, ci);
    // staging: every item exactly once, in order, inside the tile
    if (d.s_lo[0] != 0 || d.s_lo[nsl] != ns) { printf("nrm %d kind %d: staging items %d .. %d of %d\n", nrm, kind, d.s_lo[0], d.s_lo[nsl], ns); ++bad; }
    for (int g = 0; g < nsl; ++g) if (d.s_lo[g + 1] < d.s_lo[g]) { printf("nrm %d kind %d: staging not monotone at slot %d\n", nrm, kind, g); ++bad; }
    // epilogue: half A's items inside slots [0, HSLOTS), half B's inside [HSLOTS, SLOTS)
    if (!nar) {
      if (d.e_lo[0] != 0 || d.e_lo[HSLOTS] != eha || d.e_lo[SLOTS] != eha + ehb) { printf("nrm %d kind %d: epilogue split %d %d %d (want 0 %d %d)\n", nrm, kind, d.e_lo[0], d.e_lo[HSLOTS], d.e_lo[SLOTS], eha, eha + ehb); ++bad; }
      for (int g = 0; g < SLOTS; ++g) if (d.e_lo[g + 1] < d.e_lo[g]) { printf("nrm %d kind %d: epilogue not monotone at slot %d\n", nrm, kind, g); ++bad; }
      // loads: every vector once, LEAD slots (or as many as the tile has) before its first item, never behind it
      int seen[MAX_NITP] = {};
      for (int g = 0; g < SLOTS; ++g) if (d.ld[g] >= 0) {
        const int k = d.ld[g];
        ++seen[k];
        int first = 0;
        while (d.s_lo[first + 1] <= k * sv) ++first;
        if (g > first || (first - g < LEAD - NITP && first >= LEAD)) { printf("nrm %d kind %d: vector %d loaded in slot %d, first used in slot %d\n", nrm, kind, k, g, first); ++bad; }
      }
      for (int k = 0; k < NITP; ++k) if (seen[k] != 1) { printf("nrm %d kind %d: vector %d loaded %d times\n", nrm, kind, k, seen[k]); ++bad; }
    }
    // the budget
    int worst = 0, total = 0;
    for (int g = 0; g < nsl; ++g) {
      int c = 0;
      for (int m = d.s_lo[g]; m < d.s_lo[g + 1]; ++m) c += s_cost(nrm, nar, m % sv);
      for (int e = d.e_lo[g]; e < d.e_lo[g + 1]; ++e) c += e < eha ? e_cost(kind, 0, e, ci) : e_cost(kind, 1, e - eha, ci);
      worst = c > worst ? c : worst; total += c;
    }
    printf("ci %d nrm %d kind %d: %d slots, %d staging + %d epilogue items, cost %d, worst slot %d, most items per slot %d / %d\n", ci, nrm, kind, nsl, ns, eha + ehb, total, worst, d.max_s, d.max_e);
    if (worst != d.max_cost || d.max_s > MAXS || d.max_e > MAXE + 2) { printf("nrm %d kind %d: table summary wrong\n", nrm, kind); ++bad; }
  }
  return bad ? 1 : 0;
}
"""


def test_deal_tables_of_the_weight_stationary_conv():
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    src = open(SRC).read()
    a, b = src.index("namespace wst {"), src.index("}  // namespace wst")
    ns = src[a:b]
    # (the device-side accessors are not part of the table; the fixed trip counts behind them are)
    ns = ns[:ns.index("template <bool NRM, int KIND, int CI> struct DealOf")] + "constexpr int MAXS = 8, MAXE = 4;\n}\n"
    assert "constexpr int MAXS = 8, MAXE = 4;" in src
    with tempfile.TemporaryDirectory() as td:
        cpp = os.path.join(td, "deal.cpp")
        open(cpp, "w").write("#define __host__\n#define __device__\n#include <cstdio>\n" + ns + MAIN)
        exe = os.path.join(td, "deal")
        r = subprocess.run([gxx, "-std=c++17", "-O1", cpp, "-o", exe], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        r = subprocess.run([exe], capture_output=True, text=True)
        print(r.stdout)
        assert r.returncode == 0, r.stdout[-2000:]
