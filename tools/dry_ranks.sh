#!/bin/bash
# tools/dry_ranks.sh -- one rank's compute schedule at N = 2, 4, 8 without communication (bench.py --transport dry),
# plus a kernel trace of the N = 8 middle rank. MEASUREMENT ONLY: the numbers of a dry run are not results.
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/dry
mkdir -p "$OUT"
cd "$ROOT"
python3 bench.py --no-cpu-baseline --steps 30 --warmup 5 > "$OUT/n1.json"
for cfg in "2 0" "4 1" "8 3" "8 0" "8 7"; do
  set -- $cfg
  python3 bench.py --gpus $1 --dry-rank $2 --no-cpu-baseline --steps 30 --warmup 5 $EXTRA > "$OUT/n$1_r$2.json"
  echo "N=$1 rank $2 done"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace8" -- python3 "$ROOT/bench.py" --gpus 8 --dry-rank 3 --no-cpu-baseline --steps 4 --warmup 1 > "$OUT/trace8.json" 2> "$OUT/trace8.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace2" -- python3 "$ROOT/bench.py" --gpus 2 --dry-rank 0 --no-cpu-baseline --steps 4 --warmup 1 > "$OUT/trace2.json" 2> "$OUT/trace2.err"
cd "$ROOT"
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/dry/n*.json')):
    d=json.load(open(f)); print(os.path.basename(f), round(d['ms_per_step'],3), 'ms/cycle', d.get('comm_per_cycle'))
PY
