"""From a rocprofv3 --kernel-trace csv: the last launches, with stream / queue ids and start-end times relative to the first shown."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 120
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[len(rows) - n - skip:len(rows) - skip]
t0 = int(rows[0]['Start_Timestamp'])
for r in rows:
    print("%-7s q%-3s %9.1f %9.1f  %7.1f  %s" % (r['Stream_Id'], r.get('Queue_Id', '?'), (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3,
          (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Kernel_Name'].replace('_ZN3pwr', '').replace('void pwr::', '')[:60]))
