#!/bin/bash
# tools/ktimes.sh REGEX VAR v1 v2 ... -- average duration of the kernels matching REGEX (rocprofv3 --kernel-trace --stats of a short
# bench.py run) once per value of an environment switch
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
RE=$1; VAR=$2; shift 2
for v in "$@"; do
  rm -rf /tmp/kt_$v; cd /tmp && export TMPDIR=/tmp
  env $VAR=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$v -- python3 $ROOT/bench.py --no-cpu-baseline --steps 6 --warmup 1 $KT_BENCH_ARGS > /dev/null 2>&1
  echo "== $VAR=$v"
  python3 - /tmp/kt_$v "$RE" <<'PY'
import csv, glob, re, sys, collections
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if re.search(sys.argv[2], r['Kernel_Name']):
        acc[(r['Kernel_Name'].replace('void mg::(anonymous namespace)::', '')[:60], r['Grid_Size_X'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for (n, g), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    v = v[len(v) // 3:]   # skip the warm-up launches
    print(f"   {n:60s} grid {g:>9} n {len(v):3d} avg {sum(v) / len(v):8.1f} us min {min(v):8.1f}")
PY
done
