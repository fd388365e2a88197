#!/bin/bash
# tools/profile_configs.sh <round-tag> -- rocprofv3 passes (kernel trace, FETCH_SIZE, WRITE_SIZE, SQ_*) of the other BASELINE
# configurations on one GPU: config 3 (red-black), config 4's grid (1025^3 fp32) and config 5 as worded (semi + zebra).
# Summaries: python tools/summarize_prof.py gpurun_out/prof_<tag>_<cfg> <tag>_<cfg>
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r03}
bash "$ROOT/tools/profile.sh" ${TAG}_config3_rbgs --smoother rbgs
bash "$ROOT/tools/profile.sh" ${TAG}_config4_grid --grid 1025 --levels 7 --dtype f32
bash "$ROOT/tools/profile.sh" ${TAG}_config5_semi_zebra --aniso-eps 0.01 --semi 3 --levels 8 --smoother zebra
