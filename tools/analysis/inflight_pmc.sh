#!/bin/bash
# usage (GPU box, repo root): tools/analysis/inflight_pmc.sh OUTDIR "LIB_DIR LIB_DIR ..." WORKLOAD [WORKLOAD...]
# Cache and instruction counters of every kernel with THREE hosts' frames in flight (rocprofv3 --pmc, one pass per
# counter group, 12 frames through tools/analysis/inflight_pmc.py), for library builds side by side; summaries by
# tools/pmc_summary.py (per kernel, mean per launch).
OUT=$1; LIBS=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
for W in "$@"; do
  for L in $LIBS; do
    export OCRT_LIB_DIR=$L
    i=0
    for C in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
             "SQ_INSTS_SMEM SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAVES" \
             "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES" \
             "FETCH_SIZE" "WRITE_SIZE"; do
      i=$((i+1))
      timeout -k 10 180 rocprofv3 --pmc $C --output-format csv -d $R/$OUT/${W}__$L/p$i -- python3 $R/tools/analysis/inflight_pmc.py $W 3 12 > $R/$OUT/$W.$L.p$i.log 2>&1 || echo "$W $L pass $i failed"
    done
    python3 $R/tools/pmc_summary.py $R/$OUT/${W}__$L > $R/$OUT/inflight_pmc_${W}__$L.txt
    echo "== $W $L: $(grep -c mean $R/$OUT/inflight_pmc_${W}__$L.txt) counter lines"
  done
done
