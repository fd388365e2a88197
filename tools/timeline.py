#!/usr/bin/env python3
"""tools/timeline.py <kernel_trace.csv> [n] [skip] -- start offset, duration and gap of n kernels (ending `skip` before the
last one) of a rocprofv3 kernel trace."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
last = rows[-n - skip:len(rows) - skip]
t0 = int(last[0]['Start_Timestamp']); prev = None
for r in last:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev) / 1e3 if prev else 0
    name = r['Kernel_Name'].replace('void mg::(anonymous namespace)::', '').replace('mg::(anonymous namespace)::', '')
    print(f"{(s - t0) / 1e3:9.1f} dur {(e - s) / 1e3:8.1f} gap {gap:6.1f} grid {r['Grid_Size_X']:>9} wg {r['Workgroup_Size_X']:>4} {name[:90]}")
    prev = e
