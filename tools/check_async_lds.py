"""Static check of a kernel's ISA text (hipcc -S) for the hazard the inline-asm LDS reads of conv_wgrad_dma.hip invite: the compiler
believes an asm's output register is defined when the asm statement ends, but a ds_read's data arrives later -- any instruction that
touches the destination registers between the read and the s_waitcnt lgkmcnt that covers it uses (or is overwritten by) stale data.
Linear scan per kernel, LDS operations retire in order.   python tools/check_async_lds.py file.s [kernel-name-substring]"""
import re, sys

def regs(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1): out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else: out.add(int(m.group(3)))
    return out

def check(lines, name):
    pending = []      # (is_read, dest regs, line no)
    bad = 0
    for no, ln in lines:
        t = ln.strip()
        if not t or t.startswith(";") or t.startswith("."): continue
        op = t.split()[0]
        if op.startswith("ds_read") or op.startswith("ds_write") or op.startswith("ds_"):
            ops = t[len(op):].split(",")
            busy = set().union(*[d for r, d, _ in pending if r]) if pending else set()
            touched = regs(t[len(op):])
            if busy & touched:
                print("%s:%d  LDS op touches registers still in flight %s: %s" % (name, no, sorted(busy & touched), t)); bad += 1
            pending.append((op.startswith("ds_read"), regs(ops[0]) if op.startswith("ds_read") else set(), no))
            continue
        m = re.search(r"lgkmcnt\((\d+)\)", t)
        if op == "s_waitcnt":
            if m:
                n = int(m.group(1))
                while len(pending) > n: pending.pop(0)
            continue
        if op in ("s_barrier",) or op.startswith("s_cbranch") or op.startswith("s_branch") or op.startswith("s_"):
            continue
        busy = set().union(*[d for r, d, _ in pending if r]) if pending else set()
        touched = regs(t[len(op):])
        if busy & touched:
            print("%s:%d  touches registers of an LDS read still in flight %s: %s" % (name, no, sorted(busy & touched), t)); bad += 1
    return bad

src = open(sys.argv[1]).read().splitlines()
want = sys.argv[2] if len(sys.argv) > 2 else ""
cur, buf, total = None, [], 0
for i, ln in enumerate(src, 1):
    m = re.match(r"^(_Z\w+):", ln)
    if m: cur, buf = m.group(1), []
    elif cur:
        buf.append((i, ln))
        if "s_endpgm" in ln:
            if want in cur:
                b = check(buf, cur[:60]); total += b
                print(cur[:80], "->", b, "findings")
            cur = None
print("total findings", total)
