mkdir -p gpurun_out/r4n
for P in 0 0.3 0.5 0.7; do OCRT_AB_PACING=$P OCRT_AB_HOSTS=3 python3 tools/ab_variants.py lib -- bunny_1080p_ao interior_1080p_ao bunny_600_defaults --reps 1 2>&1 | sed "s/^/pacing $P: /"; done > gpurun_out/r4n/pacing.log 2>&1
for B in 1152 1408 1664 2048; do OCRT_AB_HOSTS=3 python3 tools/ab_variants.py lib_knobs:OCRT_AO_BLOCKS=$B -- bunny_1080p_ao interior_1080p_ao bunny_600_defaults --reps 1 2>&1; done > gpurun_out/r4n/blocks.log 2>&1
for K in OCRT_BATCH_BELOW=32 OCRT_BATCH_BELOW=48 OCRT_BATCH_BELOW=64 OCRT_COST_SHIFT=0 OCRT_COST_SHIFT=4; do OCRT_AB_HOSTS=3 python3 tools/ab_variants.py lib_knobs:$K -- terrain_2m_1080p_ao bunny_1080p_ao --reps 1 2>&1; done > gpurun_out/r4n/knobs_terrain.log 2>&1
cat gpurun_out/r4n/pacing.log gpurun_out/r4n/blocks.log gpurun_out/r4n/knobs_terrain.log | cut -c1-175
