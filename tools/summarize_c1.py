#!/usr/bin/env python3
"""Summarises tools/prof_c1.sh (BASELINE config 1 through the CLI) into profiles/r01_config1_cli_kernel_stats.md."""
import collections
import csv
import glob
import os

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
base = os.path.join(ROOT, "gpurun_out", "prof_c1")


def newest(pat):
    f = sorted(glob.glob(os.path.join(base, pat)), key=os.path.getmtime)
    return f[-1] if f else None


rows = list(csv.DictReader(open(newest("trace/*/*_kernel_stats.csv"))))
out = open(os.path.join(ROOT, "profiles", "r01_config1_cli_kernel_stats.md"), "w")
out.write("# rocprofv3 --kernel-trace --stats of `Multigrid -n 257 -a 1 -w 10 -ml 3 -test 1 -smt 1` "
          "(BASELINE config 1, the reference's own case)\n\n")
out.write("| kernel | calls | total ms | avg µs | % |\n|---|---|---|---|---|\n")
for r in rows[:12]:
    n = r["Name"].replace("void mg::(anonymous namespace)::", "").split("(")[0]
    out.write(f"| `{n}` | {r['Calls']} | {int(r['TotalDurationNs']) / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.1f} | {r['Percentage']} |\n")
acc = collections.defaultdict(float)
for d in ("pmc1", "pmc2"):
    f = newest(f"{d}/*/*_counter_collection.csv")
    if not f:
        continue
    for r in csv.DictReader(open(f)):
        if "coarse" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
out.write("\nSQ counters summed over the 13 coarse-solver launches (separate --pmc passes):\n\n")
for k, v in sorted(acc.items()):
    out.write(f"* {k} = {v:.0f}\n")
out.close()
print(open(os.path.join(ROOT, "profiles", "r01_config1_cli_kernel_stats.md")).read())
