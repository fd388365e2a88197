#!/usr/bin/env python3
"""tools/kres.py <file.hip> [regex] -- per-kernel register / LDS / spill table from hipcc's
-Rpass-analysis=kernel-resource-usage remarks (cross-compiles for gfx950, no GPU needed)."""
import os
import re
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def main():
    src = sys.argv[1]
    pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
    extra = sys.argv[3:]
    p = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-std=c++17",
                        "-I" + os.path.join(ROOT, "include"), "-c", src, "-o", "/dev/null",
                        "-Rpass-analysis=kernel-resource-usage", *extra], capture_output=True, text=True)
    rows, cur = [], {}
    for line in p.stderr.splitlines():
        m = re.search(r":\d+:\d+:\s+remark:\s+(.*?) \[-Rpass", line) or re.search(r":\d+:\d+:\s+(.*?) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:") or t.startswith("Name:"):
            if cur:
                rows.append(cur)
            cur = {"name": t.split(":", 1)[1].strip()}
        elif ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    if cur:
        rows.append(cur)
    names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows),
                           capture_output=True, text=True).stdout.splitlines()
    for r, name in zip(rows, names):
        name = re.sub(r"\(mg::Geom.*", "", name).replace("void mg::(anonymous namespace)::", "")
        if not pat.search(name):
            continue
        g = lambda k: r.get(k, "?")
        print(f"{name:84s} vgpr {g('VGPRs'):>4} agpr {g('AGPRs'):>3} spill {g('VGPRs Spill'):>4} scratch {g('ScratchSize [bytes/lane]'):>4} "
              f"occ {g('Occupancy [waves/SIMD]')} lds {g('LDS Size [bytes/block]')}")


if __name__ == "__main__":
    main()
