#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc_collect.sh OUTDIR WORKLOAD [WORKLOAD...]
# Collects the SQ / TCC / TCP counter passes for the ray-casting kernels of each bench workload: one rocprofv3
# --pmc run per pass (counters only, no tracing in the same run), two frames each through tools/prof_run.py, then
# folds them into OUTDIR/pmc.json (tools/pmc_to_json.py), stamped with the hash of the kernel sources.
OUT=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
for W in "$@"; do
  i=0
  for C in \
   "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
   "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
   "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
   "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE" \
   "FETCH_SIZE" "WRITE_SIZE" ; do
    i=$((i+1))
    timeout -k 10 180 rocprofv3 --pmc $C --output-format csv -d $R/$OUT/$W/p$i -- python3 $R/tools/prof_run.py --frames 2 --workload $W > $R/$OUT/$W.p$i.log 2>&1 || echo "$W pass $i failed"
  done
  python3 $R/tools/pmc_summary.py $R/$OUT/$W > $R/$OUT/pmc_$W.txt
  # the instruction counts once more for the grid a host launches when it SHARES its GPU (a ring of three: 4.5 workgroups
  # per CU instead of 8 -- other claim counts, hence other instruction counts)
  timeout -k 10 180 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $R/$OUT/${W}__shared/p1 -- python3 $R/tools/prof_run.py --frames 2 --share 3 --workload $W > $R/$OUT/$W.shared.log 2>&1 || echo "$W shared pass failed"
  python3 $R/tools/pmc_summary.py $R/$OUT/${W}__shared > $R/$OUT/pmc_${W}__shared.txt
  echo "$W: counters collected"
done
python3 $R/tools/pmc_to_json.py $R/$OUT "$@" > $R/$OUT/pmc.json
