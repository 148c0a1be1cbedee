"""Register / LDS / scratch use of the kernels in the built library whose name contains a pattern:  python tools/kernel_regs.py wgrad3"""
import os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pixelwiseregression_amd import codeobj_scan as cs
lib = os.path.join(os.path.dirname(cs.__file__), "libpwr_hip.so")
pat = sys.argv[1] if len(sys.argv) > 1 else ""
for triple, blob in cs.code_objects(lib):
    if "gfx950" not in triple:
        continue
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(blob); f.flush()
        txt = subprocess.run([os.path.join(cs.LLVM, "llvm-readelf"), "--notes", f.name], capture_output=True, text=True).stdout
    for m in re.finditer(r"- \.agpr_count:\s+(\d+).*?\.group_segment_fixed_size:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)", txt, re.S):
        ag, lds, name, scr, sg, vg = m.groups()
        if pat in name:
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            print("vgpr %3s agpr %3s sgpr %3s lds %6s scratch %4s  %s" % (vg, ag, sg, lds, scr, dem[:120]))
