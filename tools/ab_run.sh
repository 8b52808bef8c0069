#!/bin/bash
# usage (GPU box, repo root): tools/ab_run.sh [frames]   -- Mrays/s of the main workloads, one line each (A/B runs).
R=${GRAFT_REPO_ROOT:-$(pwd)}
F=${1:-5}
for W in bunny_1080p_ao bunny_1080p_primary bunny_600_defaults bunny_1080p_s16 bunny_1080p_s64 interior_1080p_ao interior_4k_ao; do
  python3 $R/tools/prof_run.py --workload $W --frames $F 2>&1 | tail -1
done
