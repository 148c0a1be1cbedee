"""Per-stream timeline of the train step from a rocprofv3 --kernel-trace CSV: which queue is the critical chain, how busy it is, where
its idle gaps are and what the side queues do meanwhile.

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d OUT -o t -- python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --accuracy-steps 0
    python tools/chain_timeline.py OUT/.../t_kernel_trace.csv
"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
name = lambda r: r.get("Kernel_Name") or r.get("Name")
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
qkey = "Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
# steady-state window: between the 6th-last and the last optimizer kernel (adamw)
opt = [r for r in rows if "adam" in name(r).lower()]
if len(opt) < 8:
    print("not enough optimizer launches", len(opt)); sys.exit(1)
t0, t1 = opt[-7]["e"], opt[-1]["e"]
nsteps = 6
win = [r for r in rows if r["s"] >= t0 and r["e"] <= t1]
byq = collections.defaultdict(list)
for r in win: byq[r[qkey]].append(r)
print("window: %d steps, %.3f ms per step, %d launches per step" % (nsteps, (t1 - t0) / nsteps / 1e6, len(win) / nsteps))
for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(r["e"] - r["s"] for r in rs)
    print("queue %s: %5.0f launches/step, busy %.3f ms/step" % (q, len(rs) / nsteps, busy / nsteps / 1e6))
chain_q = max(byq, key=lambda q: sum(1 for r in byq[q] if "conv3x3_patch" in name(r)))
ch = byq[chain_q]
gaps = collections.Counter(); gapn = collections.Counter()
for a, b in zip(ch, ch[1:]):
    g = b["s"] - a["e"]
    if g > 0:
        k = name(a)[:50] + " -> " + name(b)[:50]
        gaps[k] += g; gapn[k] += 1
tot_gap = sum(gaps.values())
print("chain queue %s: idle %.3f ms/step in %d gaps/step; largest contributors:" % (chain_q, tot_gap / nsteps / 1e6, sum(gapn.values()) / nsteps))
for k, g in gaps.most_common(12):
    print("  %7.1f us/step  x%5.1f  mean %5.1f us   %s" % (g / nsteps / 1e3, gapn[k] / nsteps, g / gapn[k] / 1e3, k))
dur = collections.Counter(); cnt = collections.Counter()
for r in ch: dur[name(r)[:70]] += r["e"] - r["s"]; cnt[name(r)[:70]] += 1
print("chain kernels by time:")
for k, d in dur.most_common(25):
    print("  %7.1f us/step  x%5.1f  mean %6.1f us   %s" % (d / nsteps / 1e3, cnt[k] / nsteps, d / cnt[k] / 1e3, k))
