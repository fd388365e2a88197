#!/bin/bash
# tools/prof_c1.sh -- rocprofv3 passes of the reference's own case (BASELINE config 1) through the CLI:
# kernel trace + SQ instruction counters for the coarse-solver kernel. Outputs in gpurun_out/prof_c1/.
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_c1
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BIN=$ROOT/multigrid_prj_amd/lib/Multigrid
ARGS="-n 257 -a 1 -w 10 -ml 3 -test 1 -smt 1"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BIN $ARGS > "$OUT/trace.out" 2> "$OUT/trace.err"
echo "trace done"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT/pmc1" -- $BIN $ARGS > "$OUT/pmc1.out" 2> "$OUT/pmc1.err"
echo "pmc1 done"
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY --output-format csv -d "$OUT/pmc2" -- $BIN $ARGS > "$OUT/pmc2.out" 2> "$OUT/pmc2.err"
echo "pmc2 done"
