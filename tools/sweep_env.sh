#!/bin/bash
# tools/sweep_env.sh VAR v1 v2 ... -- bench.py once per value of one environment switch (other settings: $FIXED), printing the
# cycle time and the finest-level launch times; no parity tests (run tools/ab.sh on the value that wins)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/sweep
VAR=$1; shift
for v in "$@"; do
  env $FIXED $VAR=$v python3 bench.py --steps ${AB_STEPS:-20} --warmup 3 --no-cpu-baseline ${AB_BENCH_ARGS} > gpurun_out/sweep/${VAR}_$v.json 2> gpurun_out/sweep/${VAR}_$v.err || { tail -3 gpurun_out/sweep/${VAR}_$v.err; continue; }
  python3 - "$VAR=$v" gpurun_out/sweep/${VAR}_$v.json <<'PY'
import json, sys
o = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][0])
k = {x["kernel"][:14]: x["launch_ms"] for x in o["kernels"]}
print(f"{sys.argv[1]:24s} ms/cycle {o['ms_per_step']:.4f}  pair {o['roofline']['launch_ms']:.4f}  " + "  ".join(f"{n}: {v:.4f}" for n, v in k.items()), flush=True)
PY
done
