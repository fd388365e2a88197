#!/usr/bin/env python3
"""tools/timeline.py <kernel_trace.csv> [marker-substring [period-index]] -- one cycle of a rocprofv3 kernel trace as a timeline.

Cuts the trace at the launches whose name contains the marker (default: the coarse solver, one per V-cycle), takes the
last complete cycle and prints every launch in start order with the idle gap before it, the stream it ran on and
whether it overlapped the previous one; then the sums: busy time (union of intervals), idle time, time per kernel."""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    marker = sys.argv[2] if len(sys.argv) > 2 else "k_coarse_"
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            wgs = int(r.get("Grid_Size_X", 0) or 0) // max(int(r.get("Workgroup_Size_X", 1) or 1), 1)
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "?")), wgs))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if marker in r[2]]
    if len(marks) < 3:
        sys.exit(f"fewer than 3 launches match {marker!r}")
    k = int(sys.argv[3]) if len(sys.argv) > 3 else len(marks) - 3   # which period (bench.py: warm-up cycles first, then the timed ones, then mg_solve's)
    a, b = marks[k], marks[k + 1]
    cyc = rows[a + 1:b + 1]
    t0 = rows[a][1]
    end_prev = t0
    busy = 0
    per = defaultdict(lambda: [0, 0.0])
    print(f"cycle of {len(cyc)} launches, {(cyc[-1][1] - t0) / 1e3:.1f} us from the end of one coarse solve to the end of the next")
    print(f"{'start us':>9} {'gap us':>7} {'dur us':>8} stream  {'wgs':>6}  kernel")
    for s, e, name, q, wgs in cyc:
        gap = (s - end_prev) / 1e3
        short = name.replace("void ", "").replace("(anonymous namespace)::", "").replace("mg::", "").split("(")[0][:80]
        print(f"{(s - t0) / 1e3:9.1f} {gap:7.1f} {(e - s) / 1e3:8.1f} {q:>6}  {wgs:6d}  {short}")
        busy += max(0, e - max(s, end_prev))
        end_prev = max(end_prev, e)
        per[short][0] += 1
        per[short][1] += (e - s) / 1e3
    total = (cyc[-1][1] - t0) / 1e3
    print(f"busy {busy / 1e3:.1f} us, idle {total - busy / 1e3:.1f} us of {total:.1f}")
    for k, (n, us) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print(f"{us:8.1f} us {n:3d} x {k}")


if __name__ == "__main__":
    main()
