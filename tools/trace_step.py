"""Summarise one train step of a rocprofv3 --kernel-trace csv: per-stream busy time and the top kernels of each stream."""
import csv, collections, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 14
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adamw' in r['Kernel_Name']]
a, b = idx[5], idx[6]
step = rows[a + 1:b + 1]
t0 = int(step[0]['Start_Timestamp'])
print(len(step), "launches,", (int(step[-1]['End_Timestamp']) - t0) / 1e6, "ms")
for sid in sorted(set(r['Stream_Id'] for r in step)):
    d = collections.defaultdict(list)
    first = last = None
    for r in step:
        if r['Stream_Id'] == sid:
            n = r['Kernel_Name'].replace('_ZN3pwr', '').replace('void pwr::', '')[:52]
            d[n].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000)
            if first is None: first = (int(r['Start_Timestamp']) - t0) / 1e6
            last = (int(r['End_Timestamp']) - t0) / 1e6
    print("stream", sid, "first %.2f last %.2f busy %.2f ms" % (first, last, sum(sum(v) for v in d.values()) / 1000))
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:top]:
        print("  %-54s %4d %8.1f us total %6.1f avg" % (k, len(v), sum(v), sum(v) / len(v)))
