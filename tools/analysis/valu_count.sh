#!/bin/bash
# usage (GPU box, repo root): tools/analysis/valu_count.sh OUTDIR "LIB_DIR LIB_DIR ..." WORKLOAD [WORKLOAD...]
# Vector / scalar instruction counts and wave cycles of each kernel for library builds side by side (one rocprofv3
# --pmc pass per build and workload, two frames through tools/prof_run.py), printed by tools/pmc_summary.py.
OUT=$1; LIBS=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
for W in "$@"; do
  for L in $LIBS; do
    export OCRT_LIB_DIR=$L
    timeout -k 10 180 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv \
      -d $R/$OUT/${W}__$L/p1 -- python3 $R/tools/prof_run.py --frames 2 --workload $W > $R/$OUT/$W.$L.log 2>&1 || echo "$W $L failed"
    echo "== $W $L"
    python3 $R/tools/pmc_summary.py $R/$OUT/${W}__$L | tee $R/$OUT/pmc_${W}__$L.txt
  done
done
