#!/bin/bash
# tools/cli_times.sh -- BASELINE config 1 and the reference's fixture run through the `Multigrid` executable: the solve timer
# with the warm-up cycle in the initialisation phase (default) and without it (-cold: the first iteration is cold, like the reference's)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
python -c "from multigrid_prj_amd import build as b; print(b.build_cli())"
BIN=$ROOT/multigrid_prj_amd/lib/Multigrid
T=$(mktemp -d); cd $T
for args in "-n 257 -a 1 -w 10 -ml 3 -test 1 -smt 1" "-n 257 -a 1 -w 10 -ml 3 -test 1 -smt 0" "-n 385 -a 1 -w 10 -ml 5 -test 0 -smt 2"; do
  for mode in "" "-cold"; do
    for rep in 1 2 3; do
      $BIN $args $mode > out.txt
      echo "$args $mode : $(grep -E 'Initialization time|Solving elapsed' out.txt | tr '\n' ' ')"
    done
  done
done
