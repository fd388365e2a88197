#!/bin/bash
# tools/zc_full.sh -- z-chunk length of the fused kernels on the whole 513^3 grid (one GPU): MG_J2_ZC forced
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/zcfull
for zc in 0 24 26 29 31 35 37 40 47; do
  MG_J2_ZC=$zc python3 bench.py --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/zcfull/zc$zc.json
  python3 - "$zc" <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/zcfull/zc{sys.argv[1]}.json'))
k={x['kernel'][:4]:x['launch_ms'] for x in d['kernels']}
print(f"zc={sys.argv[1]:>2} cycle {d['ms_per_step']:.3f} ms  pair {d['roofline']['launch_ms']*1e3:.1f} us  corr/rr {[round(v*1e3,1) for v in k.values()]}")
PY
done
