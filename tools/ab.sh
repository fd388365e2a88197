#!/bin/bash
# tools/ab.sh "<ENV=.. ENV=..>" ["<ENV..>" ...] -- A/B runs on the GPU box: for every environment setting the
# parity tests that exercise the fused finest-level kernels, then a short bench line; results in gpurun_out/ab_*.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
i=0
for setting in "$@"; do
    i=$((i + 1))
    echo "=== [$i] $setting"
    env $setting timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -x --timeout 280 -k "${AB_TESTS:-vcycle_extension or headline}" > gpurun_out/ab_test_$i.log 2>&1
    rc=$?
    tail -1 gpurun_out/ab_test_$i.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: stopping"; exit 1; fi
    env $setting python bench.py --steps ${AB_STEPS:-20} --warmup 3 --no-cpu-baseline ${AB_BENCH_ARGS} > gpurun_out/ab_bench_$i.json 2> gpurun_out/ab_bench_$i.err || { tail -5 gpurun_out/ab_bench_$i.err; continue; }
    python - "$setting" gpurun_out/ab_bench_$i.json <<'PY'
import json, sys
o = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][0])
k = {x["kernel"][:24]: x["launch_ms"] for x in o["kernels"]}
print(f"   {sys.argv[1]:40s} ms/cycle {o['ms_per_step']:.4f}  pair {o['roofline']['launch_ms']:.4f}  " + "  ".join(f"{n}: {v:.4f}" for n, v in k.items()))
PY
done
