#!/bin/bash
# tools/dry_zc.sh -- z-chunk length of the fused pair on thin slabs: one rank's schedule (dry run, no communication)
# at N = 8 / 4 / 2 with MG_J2_ZC forced; prints the profiled smoother segment (interior + boundary launches) per pair.
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/dryzc
for cfg in "8 3" "4 1" "2 0"; do
  set -- $cfg
  for zc in 0 4 6 8 10 12 15 16 20 24 31 32; do
    MG_J2_ZC=$zc python3 bench.py --gpus $1 --dry-rank $2 --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/dryzc/n$1_zc$zc.json
    python3 - "$1" "$zc" <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/dryzc/n{sys.argv[1]}_zc{sys.argv[2]}.json'))
print(f"N={sys.argv[1]} zc={sys.argv[2]:>2} cycle {d['ms_per_step']:.3f} ms  pair segment {d['roofline']['launch_ms']*1e3:.1f} us")
PY
  done
done
