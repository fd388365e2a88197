#!/bin/bash
# tools/dry_sweep.sh N RANK VAR v1 v2 ... -- one rank's schedule (dry run, no communication) once per value of an environment switch
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/sweep
N=$1; R=$2; VAR=$3; shift 3
for v in "$@"; do
  env $FIXED $VAR=$v python3 bench.py --gpus $N --dry-rank $R --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/sweep/dry${N}_${VAR}_$v.json 2> gpurun_out/sweep/dry${N}_${VAR}_$v.err || { tail -3 gpurun_out/sweep/dry${N}_${VAR}_$v.err; continue; }
  python3 - "N=$N $VAR=$v" gpurun_out/sweep/dry${N}_${VAR}_$v.json <<'PY'
import json, sys
o = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][0])
k = {x["kernel"][:14]: x["launch_ms"] for x in o["kernels"]}
print(f"{sys.argv[1]:28s} ms/cycle {o['ms_per_step']:.4f}  pair {o['roofline']['launch_ms']:.4f}  " + "  ".join(f"{n}: {v:.4f}" for n, v in k.items()), flush=True)
PY
done
