#!/bin/bash
# usage (GPU box, repo root): tools/bench_ab.sh "ENV=VALUE ..." ["ENV=VALUE ..." ...]  -- bench.py value / ms per step of the main
# workloads under each environment setting (one line per setting and workload), twice, for A/B runs within one box.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
  for SETTING in "$@"; do
    for W in bunny_1080p_ao bunny_600_defaults bunny_1080p_s16 interior_1080p_ao interior_4k_ao; do
      echo -n "[$SETTING] $W: "
      env $SETTING python3 $R/bench.py --no-cpu-baseline --workload $W --steps 30 2>/dev/null | tail -1 | python3 -c "import sys,json; b=json.loads(sys.stdin.read()); print(b['value'], b['ms_per_step'])"
    done
  done
done
