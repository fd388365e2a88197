#!/bin/bash
# tools/bench_configs.sh <tag> -- the other BASELINE configurations / extensions on one GPU, one JSON line each under gpurun_out/
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
TAG=${1:-r02}
run() { name=$1; shift; python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/${TAG}_bench_$name.json 2> gpurun_out/${TAG}_bench_$name.err || tail -3 gpurun_out/${TAG}_bench_$name.err
  python - gpurun_out/${TAG}_bench_$name.json $name <<'PY'
import json, sys
try:
    o = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
    print(f"{sys.argv[2]:28s} {o['ms_per_step']:8.3f} ms/cycle  drop {o['residual_drop_per_cycle']:.3f}  smoother launch {o['roofline']['launch_ms']:.3f} ms x{o['roofline']['sweeps_per_launch']:.0f} sweeps ({o['roofline']['achieved']:.0f} GB/s compulsory)")
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
run config2 --grid 257 --levels 5
run config3_rbgs --smoother rbgs
run config4_grid_1gpu --grid 1025 --levels 7 --dtype f32
run config5_semi_jacobi --aniso-eps 0.01 --semi 3 --levels 8
run config5_semi_rbgs --aniso-eps 0.01 --semi 3 --levels 8 --smoother rbgs
run config5_semi_zebra --aniso-eps 0.01 --semi 3 --levels 8 --smoother zebra
run zebra_y_aniso100 --smoother zebra --aniso-y 100
run zebra_x_aniso100 --smoother zebrax --aniso-x 100
run grid385_f64 --grid 385 --levels 5
run grid1025_f64 --grid 1025 --levels 7
