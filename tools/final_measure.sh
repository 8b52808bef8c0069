#!/bin/bash
# usage (GPU box, repo root): tools/final_measure.sh OUTDIR  -- everything the round's profiles/ directory is made of:
# PMC passes per workload (-> pmc.json), rocprofv3 --kernel-trace --stats of the default bench command and of the same
# command with one frame at a time, and the bench lines of every workload (the PMC json must be in place first so that
# the lines carry the roofline).
OUT=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$OUT
WORKLOADS="bunny_1080p_ao bunny_1080p_primary bunny_600_defaults bunny_1080p_s64 interior_1080p_ao interior_4k_ao interior_hard_1080p_ao interior_hard_4k_ao terrain_2m_1080p_ao terrain_20m_1080p_ao"
WORKLOADS=${OCRT_MEASURE_WORKLOADS:-$WORKLOADS}   # (a subset, for a call that must fit the box's time limit)
STAGE=${2:-all}   # pmc | bench | all  (a gpurun call lasts 20 minutes at most: the two stages fit one each; between them
                  # profiles/pmc.json has to be installed -- tools/install_profiles.py -- because the bench lines read it)
if [ $STAGE = pmc ] || [ $STAGE = all ]; then
  $R/tools/pmc_collect.sh $OUT/pmc $WORKLOADS > $R/$OUT/pmc.log 2>&1
  cp $R/$OUT/pmc/pmc.json $R/profiles/pmc.json
  echo "pmc stage done: $(ls $R/$OUT/pmc/pmc_*.txt | wc -l) summaries"
fi
if [ $STAGE = stats ] || [ $STAGE = all ]; then
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end > $R/$OUT/stats_bench.log 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats_one -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --in-flight 1 > $R/$OUT/stats_one_bench.log 2>&1
  cd $R
  echo "stats stage done"
fi
if [ $STAGE = bench ] || [ $STAGE = all ]; then
  cd $R
  for W in $WORKLOADS; do
    python3 bench.py --steps 20 --warmup 5 --workload $W $( [ $W = bunny_1080p_ao ] || echo --no-cpu-baseline ) 2>/dev/null | tail -1 > $OUT/bench_$W.json
    echo "$W: $(cut -c1-160 $OUT/bench_$W.json)"
  done
fi
