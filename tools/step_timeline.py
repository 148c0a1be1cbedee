"""One train step's kernel timeline out of a rocprofv3 --kernel-trace run of bench.py (run on the GPU box right after the trace):
   python tools/step_timeline.py <dir with *_kernel_trace.csv> <out.tsv> [step index from the end = 5]
Rows: queue, start offset (us, from the end of the previous step's adamw_kernel), duration (us), grid workgroups, short kernel name.
A step = the launches between two consecutive adamw_kernel ends.  Also prints, per queue, busy time and the last end."""
import collections, csv, glob, os, sys

src, dst = sys.argv[1], sys.argv[2]
back = int(sys.argv[3]) if len(sys.argv) > 3 else 5
f = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    wg = max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
    grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // wg
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"], grid))
rows.sort()
ends = [e for (s, e, q, n, g) in rows if "adamw_kernel" in n]
# consecutive optimizer launches of the timed loop are one step apart; take a pair well inside it
pairs = [(a, b) for a, b in zip(ends, ends[1:]) if b - a < 20e6]
a, b = pairs[-back]


def short(n):
    n = n.replace("void pwr::", "").replace("pwr::", "")
    return n[:n.find("(")] if "(" in n and not n.startswith("(") else n[:90]


sel = [r for r in rows if r[0] >= a - 2000 and r[1] <= b]
sel = [r for r in sel if r[1] > a]
qs = sorted({r[2] for r in sel})
with open(dst, "w") as o:
    o.write("# step of %.1f us; %d launches; queues %s\n" % ((b - a) / 1e3, len(sel), qs))
    for (s, e, q, n, g) in sel:
        o.write("%s\t%.1f\t%.1f\t%d\t%s\n" % (qs.index(q), (s - a) / 1e3, (e - s) / 1e3, g, short(n)))
print("step %.1f us, %d launches" % ((b - a) / 1e3, len(sel)))
for q in qs:
    rq = [r for r in sel if r[2] == q]
    print("queue", qs.index(q), "launches", len(rq), "busy %.0f us" % (sum(e - s for s, e, *_ in rq) / 1e3),
          "first start %.0f last end %.0f" % ((min(r[0] for r in rq) - a) / 1e3, (max(r[1] for r in rq) - a) / 1e3))
