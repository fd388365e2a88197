#!/bin/bash
# tools/profile.sh <tag> [bench args...] -- rocprofv3 passes of bench.py on the GPU box.
#   1. --kernel-trace --stats      -> per-kernel time table
#   2. --pmc FETCH_SIZE            -> HBM-side read traffic   (own pass, gfx950: x2 correction)
#   3. --pmc WRITE_SIZE            -> HBM-side write traffic  (own pass)
#   4. --pmc SQ_* (8 counters)     -> wave-cycle breakdown    (own pass)
# Outputs land in gpurun_out/prof_<tag>/ ; summarise with tools/summarize_prof.py.
set -eo pipefail
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 4 --warmup 1 --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_trace.json" 2> "$OUT/trace.err"
echo "trace pass done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline $* > "$OUT/bench_fetch.json" 2> "$OUT/fetch.err"
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline $* > "$OUT/bench_write.json" 2> "$OUT/write.err"
echo "write pass done"
# SQ pass (8 slots): where the waves' cycles go -- parked (WAIT_ANY), issue-stalled (WAIT_INST_ANY), issuing (ACTIVE_INST_*)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d "$OUT/pmc_sq" -- python3 "$ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline $* > "$OUT/bench_sq.json" 2> "$OUT/sq.err"
echo "sq pass done"
find "$OUT" -name "*.csv" | head -20
