"""In-tree build of libpwr_hip.so (gfx950) with hipcc.  No torch C++ ABI involved: the library is a
plain C ABI (include/pwr.h) loaded with ctypes, which side-steps the hipcc 7.2 / torch-HIP 7.0 skew.

    python -m pixelwiseregression_amd.build [--force] [--debug]
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libpwr_hip.so")
OBJ = os.path.join(HERE, "csrc", "_obj")
ARCH = "gfx950"
# -fno-slp-vectorize: the SLP vectoriser packs adjacent scalar f32 adds / muls into v_pk_add_f32 / v_pk_mul_f32 with op_sel
# operand selection (e.g. "v_pk_add_f32 v[24:25], v[126:127], v[128:129] op_sel:[0,1]": low result = src0.lo + src1.HI).  In
# decode_bwd_cached exactly that form produced, about once per 7000 train steps on some MI355X boxes and only while MFMA
# weight-gradient kernels of another stream shared the CU, a low-half result in which the src1.hi addend was missing for lanes
# 48-63 (six captures).  Without the vectoriser no cross-half op_sel form is emitted anywhere in the library
# (tests/test_boundary_cpu.py checks the shipped code objects), the events are gone on the boxes that showed them, and the train
# step takes the same time (packed f32 math beside MFMAs is an anti-lever on this chip anyway, MI355X_MICROARCH.md).
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
         "-Wno-unused-result", "-ffp-contract=off", "-fno-slp-vectorize"]


# per-source flags.  conv_wstat.hip: its K loop (36 steps x 8 slots, every register index a compile-time constant) must unroll completely;
# the default cap on a pragma-requested unroll (16384 IR instructions, counted BEFORE the per-slot constants fold) stops it silently otherwise
# (a warning, and a kernel whose register arrays live in scratch)
# -amdgpu-mfma-vgpr-form: its accumulators live in VGPRs (the epilogue reads them with vector instructions; the AGPR file holds weights)
PER_FILE_FLAGS = {"conv_wstat.hip": ["-mllvm", "-pragma-unroll-threshold=1000000", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-Werror=pass-failed"]}
# experiment KERNELS that only the debug build compiles and links (never part of the product).  Standalone probes with a main() of their
# own live in tools/probes/ and are built by their scripts (tools/issue_probe.sh), not into any library.
DEBUG_SRC = os.path.join(ROOT, "tools", "csrc_debug")
_debug_sources = False


def _sources():
    """source paths: csrc/*.hip, and in the debug build tools/csrc_debug/*.hip as well"""
    out = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".cpp"))]
    if _debug_sources and os.path.isdir(DEBUG_SRC):
        out += [os.path.join(DEBUG_SRC, f) for f in sorted(os.listdir(DEBUG_SRC)) if f.endswith(".hip")]
    return out


def _digest(paths):
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=True, debug=False, extra_flags=()):
    """Compile csrc/*.hip for gfx950 and link libpwr_hip.so (in-tree, next to this file).

    debug=True builds the DEBUG variant instead: -DPWR_DEBUG_BUILD (experiment switches read PWR_* environment variables, the
    debugging entry points of include/pwr_debug.h exist), into tools/_build/libpwr_hip_dbg.so -- never into the package directory;
    tools/dbglib.py loads it for the measurement scripts.  extra_flags: further -D / -f flags for a debug variant.
    Every freshly linked library is scanned for the packed-f32 instruction form of DESIGN.md section 2 (codeobj_scan)."""
    global FLAGS, LIB, OBJ, _debug_sources
    if debug or extra_flags:
        if not debug:
            raise ValueError("extra_flags only with debug=True: the product library has one configuration")
        saved = (FLAGS, LIB, OBJ)
        bdir = os.path.join(ROOT, "tools", "_build")
        tag = "".join(c if c.isalnum() else "_" for c in "".join(extra_flags))        # a variant per set of extra flags
        FLAGS, LIB, OBJ = (FLAGS + ["-DPWR_DEBUG_BUILD"] + list(extra_flags), os.path.join(bdir, "libpwr_hip_dbg%s.so" % tag),
                           os.path.join(bdir, "obj" + tag))
        _debug_sources = True
        try:
            return build(force, verbose)
        finally:
            FLAGS, LIB, OBJ = saved
            _debug_sources = False
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".h", ".inc"))]
    headers += [os.path.join(ROOT, "include", "pwr.h"), os.path.join(ROOT, "include", "pwr_debug.h")]
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    jobs = []
    for sp in _sources():
        op = os.path.join(OBJ, os.path.basename(sp) + ".o")
        stamp = op + ".sha"
        dig = _digest([sp] + headers) + "|" + " ".join(PER_FILE_FLAGS.get(os.path.basename(sp), []))
        if not force and os.path.exists(op) and os.path.exists(stamp) and open(stamp).read() == dig:
            continue
        jobs.append((sp, op, stamp, dig))

    def run(job):
        sp, op, stamp, dig = job
        cmd = [hipcc] + FLAGS + PER_FILE_FLAGS.get(os.path.basename(sp), []) + ["-x", "hip", "-c", sp, "-o", op]
        if verbose:
            print("[pwr build]", os.path.basename(sp), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (sp, r.stdout, r.stderr))
        with open(stamp, "w") as f:
            f.write(dig)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, os.path.basename(s_) + ".o") for s_ in _sources()]
    if jobs or not os.path.exists(LIB) or force:
        cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("[pwr build] linked", LIB, flush=True)
        _scan_gate(LIB, verbose)
    else:
        from . import codeobj_scan
        import json
        ok = False
        try:
            rec = json.load(open(codeobj_scan.stamp_path(LIB)))
            ok = rec.get("sha256") == codeobj_scan.lib_digest(LIB)
        except (OSError, ValueError):
            pass
        if not ok:                       # an up-to-date library without a matching scan record: scan it now (never delete it for a
            _scan_gate(LIB, verbose, keep=True)      # missing toolchain: tests/test_boundary_cpu.py flags a library without a record)
    return LIB


def _scan_gate(lib, verbose=True, keep=False):
    """No packed f32 instruction with a cross-half op_sel may ship, and no instruction may touch the
    destination of an inline-asm LDS read before its wait (conv_wgrad_dma.hip, the loader waves of conv_wgrad_ws.hip): checked on every
    link, with the LLVM tools of the hipcc that built the library.  The result is written beside the library (<lib>.scan.json, keyed by
    the library's sha256): tests/test_boundary_cpu.py refuses a product library without a matching, clean record.  A toolchain without
    llvm-objdump FAILS the product build (the gate is the only protection against those two instruction forms, and a foreign toolchain is
    exactly where they would come back) unless PWR_ALLOW_UNSCANNED=1 is set -- the record then says "unscanned"."""
    import json
    from . import codeobj_scan
    if not codeobj_scan.available():
        if os.environ.get("PWR_ALLOW_UNSCANNED") != "1":
            if not keep:
                os.remove(lib)
            raise RuntimeError("llvm-objdump / llvm-objcopy not found next to %s: the code-object scan cannot run.  Point HIPCC at a ROCm "
                               "prefix that has lib/llvm/bin, or set PWR_ALLOW_UNSCANNED=1 to build an UNSCANNED library (tests will flag it)"
                               % os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"))
        with open(codeobj_scan.stamp_path(lib), "w") as f:
            json.dump({"sha256": codeobj_scan.lib_digest(lib), "scanned": False}, f)
        if verbose:
            print("[pwr build] llvm-objdump not found: code-object scan SKIPPED by PWR_ALLOW_UNSCANNED=1", flush=True)
        return
    r = codeobj_scan.scan(lib)
    if r["packed_f32_cross_half_op_sel"]:
        os.remove(lib)
        raise RuntimeError("%s contains %d packed f32 instructions with a cross-half op_sel (e.g. %s): the build flags lost "
                           "-fno-slp-vectorize or hand-written packed math was added -- see DESIGN.md section 2"
                           % (lib, r["packed_f32_cross_half_op_sel"], r["examples"][:2]))
    if r["async_lds_hazards"]:
        os.remove(lib)
        raise RuntimeError("%s: %d instructions touch the destination of an inline-asm LDS read before the s_waitcnt that covers it "
                           "(e.g. %s) -- see codeobj_scan.async_lds_hazards" % (lib, r["async_lds_hazards"], r["async_lds_examples"][:2]))
    with open(codeobj_scan.stamp_path(lib), "w") as f:
        json.dump({"sha256": codeobj_scan.lib_digest(lib), "scanned": True, "llvm": codeobj_scan.LLVM, "functions": r["functions"],
                   "packed_f32": r["packed_f32"], "packed_f32_cross_half_op_sel": 0, "async_lds_hazards": 0}, f)
    if verbose:
        print("[pwr build] code-object scan ok: %d functions, %d packed f32, 0 with cross-half op_sel, 0 uses of an LDS read in flight"
              % (r["functions"], r["packed_f32"]), flush=True)


if __name__ == "__main__":
    build(force="--force" in sys.argv, debug="--debug" in sys.argv)
