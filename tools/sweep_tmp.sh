cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc3
for C in "SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES_DUPLICATE"; do
  i=$((i+1))
  timeout -k 10 180 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc3/p$i -- python3 $R/tools/prof_run.py --frames 2 --workload bunny_1080p_ao > $R/gpurun_out/pmc3.p$i.log 2>&1 || echo "pass $i failed"
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc3 kernel
