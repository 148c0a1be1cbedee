"""Disassemble the gfx950 code objects embedded in libpwr_hip.so and report instruction forms.

    python -m pixelwiseregression_amd.codeobj_scan [lib.so]         # prints counts of packed-f32 forms

The static gate of the round-2 reproducibility fix: the shipped library must contain no packed f32 instruction whose LOW result
selects a source's HIGH register (op_sel:[..1..]) -- the instruction form that was caught producing a wrong addend in lanes 48-63
(DESIGN.md section 2).  build.build() runs it on every freshly linked library (a flag change cannot ship silently) and
tests/test_boundary_cpu.py on the library that is loaded.
"""
import os, re, struct, subprocess, sys, tempfile

def _llvm_dir():
    """The LLVM tools of the toolchain that BUILDS the library: next to the hipcc in use (HIPCC, default /opt/rocm/bin/hipcc:
    <prefix>/bin/hipcc -> <prefix>/lib/llvm/bin), so that a library built with another ROCm prefix is scanned with that prefix's
    disassembler -- not skipped because /opt/rocm happens to be absent."""
    hipcc = os.path.realpath(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"))
    for cand in (os.path.join(os.path.dirname(os.path.dirname(hipcc)), "lib", "llvm", "bin"), os.path.join(os.path.dirname(hipcc)), "/opt/rocm/lib/llvm/bin"):
        if all(os.path.exists(os.path.join(cand, t)) for t in ("llvm-objcopy", "llvm-objdump")):
            return cand
    return None


LLVM = _llvm_dir() or "/opt/rocm/lib/llvm/bin"


def available():
    return all(os.path.exists(os.path.join(LLVM, t)) for t in ("llvm-objcopy", "llvm-objdump"))


def stamp_path(lib):
    return lib + ".scan.json"


def lib_digest(lib):
    import hashlib
    h = hashlib.sha256()
    with open(lib, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib):
    """Yields (triple, bytes) of every device code object in the library's .hip_fatbin section."""
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib], check=True)
        data = open(fat, "rb").read()
    pos = data.find(MAGIC)
    while pos >= 0:
        n = struct.unpack_from("<Q", data, pos + len(MAGIC))[0]
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, q)
            triple = data[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if size:
                yield triple, data[pos + off:pos + off + size]
        pos = data.find(MAGIC, pos + len(MAGIC))


def disassemble(lib):
    """Concatenated llvm-objdump -d text of all gfx950 code objects."""
    out = []
    for triple, blob in code_objects(lib):
        if "gfx950" not in triple:
            continue
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(blob); f.flush()
            out.append(subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", f.name], capture_output=True, text=True,
                                      check=True).stdout)
    return "\n".join(out)


PK = re.compile(r"\bv_pk_(add|mul|fma)_f32\b")
# op_sel:[a,b(,c)] selects, for the LOW result, which register of each source pair is read; any 1 = a high register feeds the low result
CROSS = re.compile(r"op_sel:\[[01,]*1[01,]*\]")


def scan(lib):
    txt = disassemble(lib)
    n_kernels = len(re.findall(r"^[0-9a-f]+ <[^>]+>:", txt, flags=re.M))
    pk = [l for l in txt.splitlines() if PK.search(l)]
    cross = [l for l in pk if CROSS.search(l)]
    hz = async_lds_hazards(txt)
    return {"functions": n_kernels, "instructions": txt.count("\n"), "packed_f32": len(pk), "packed_f32_cross_half_op_sel": len(cross),
            "examples": [c.split("//")[0].strip() for c in cross[:5]], "async_lds_hazards": len(hz), "async_lds_examples": hz[:5]}


def _vregs(text):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", text):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def async_lds_hazards(txt, name_filter=("wgrad3d", "wgrad3w")):
    """The kernels of csrc/conv_wgrad_dma.hip and the loader waves of csrc/conv_wgrad_ws.hip read LDS through inline asm (so that the compiler's wait-count pass does not fence the
    LDS-DMA prefetches).  The compiler then believes an asm's output register is defined when the statement ends, while the data of
    a ds_read arrives later: any instruction that touches such a register between the read and the `s_waitcnt lgkmcnt` covering it
    copies or clobbers stale data -- round 3 shipped-candidate build had exactly that (a v_mov the register allocator put in front
    of a wait where two paths merged; wrong weight gradients in roughly one launch of ten).  Linear scan per kernel, LDS operations
    retire in order.  Returns the list of offending instruction lines."""
    bad, cur, pending = [], None, []
    for ln in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", ln)
        if m:
            cur, pending = (m.group(1) if any(f in m.group(1) for f in name_filter) else None), []
            continue
        if cur is None:
            continue
        t = ln.split("//")[0].strip()
        if not t:
            continue
        op = t.split()[0]
        rest = t[len(op):]
        busy = set().union(*[d for d in pending]) if pending else set()
        if op.startswith("ds_"):
            if busy & _vregs(rest):
                bad.append(cur[:48] + ": " + t)
            pending.append(_vregs(rest.split(",")[0]) if op.startswith("ds_read") else set())
            continue
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", t)
            if m:
                while len(pending) > int(m.group(1)):
                    pending.pop(0)
            continue
        if op.startswith("s_"):
            continue
        if busy & _vregs(rest):
            bad.append(cur[:48] + ": " + t)
    return bad


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "pixelwiseregression_amd", "libpwr_hip.so")
    print(lib, scan(lib))
