#!/usr/bin/env python3
"""Print kernel-stats and a window of the kernel timeline from a rocprofv3 --kernel-trace --stats csv directory."""
import csv
import glob
import sys

d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
f = glob.glob(d + '/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r['Name'][:80].ljust(80), r['Calls'], r['AverageNs'], r['Percentage'])
t = glob.glob(d + '/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(t)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'deepfm_fwd_bwd' in r['Kernel_Name']]
mid = idx[len(idx) // 2]
t0 = int(rows[mid]['Start_Timestamp'])
for r in rows[mid - 2:mid + n]:
    print('%8.2f %8.2f  q=%s %s' % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3,
                                   r['Queue_Id'], r['Kernel_Name'][:60]))
