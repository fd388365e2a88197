#!/bin/bash
# tools/dry_c4.sh -- BASELINE config 4's grid (1025^3 fp32, 7 levels): one rank's compute schedule at N = 8 / 4 / 2 (dry run)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/sweep
for cfg in "8 3" "4 1" "2 0"; do
  set -- $cfg
  for v in 1 0; do
    MG_FUSED_PROLONG_SLAB=$v python3 bench.py --grid 1025 --levels 7 --dtype f32 --gpus $1 --dry-rank $2 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/sweep/c4_n$1_fold$v.json 2> gpurun_out/sweep/c4_n$1_fold$v.err || { tail -3 gpurun_out/sweep/c4_n$1_fold$v.err; continue; }
    python3 - "N=$1 fold=$v" gpurun_out/sweep/c4_n$1_fold$v.json <<'PY'
import json, sys
o = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][0])
k = {x["kernel"][:14]: x["launch_ms"] for x in o["kernels"]}
print(f"{sys.argv[1]:16s} ms/cycle {o['ms_per_step']:.4f}  pair {o['roofline']['launch_ms']:.4f}  " + "  ".join(f"{n}: {v:.4f}" for n, v in k.items()), o['comm_per_cycle'], flush=True)
PY
  done
done
