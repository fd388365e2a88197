#!/bin/bash
# tools/sweep_wide.sh -- z-chunk lengths of the wide-tile kernels on the default workload, one bench line each (same box)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
show() { python -c "
import json,sys
o=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(sys.argv[1], round(o['ms_per_step'],4), 'pair', round(o['roofline']['launch_ms'],4), [round(k['launch_ms'],4) for k in o['kernels']])" "$1"; }
for z in 0 8 12 16 24 32 43 64; do MG_RRW_ZCC=$z python bench.py --no-cpu-baseline --steps 20 2>/dev/null | show "rrw_zcc=$z"; done
for z in 0 16 24 32 47 64 86 128 171 257; do MG_PW_ZC=$z python bench.py --no-cpu-baseline --steps 20 2>/dev/null | show "pw_zc=$z"; done
