#!/bin/bash
# usage (GPU box, repo root): tools/analysis/strip_sweep.sh OUTDIR  -- ms per frame by strip width (A/B build: OCRT_STRIP_TILES)
OUT=$1; mkdir -p $OUT
for S in 2 4 8 16 32; do
  OCRT_AB_HOSTS=3 python3 tools/ab_variants.py lib_knobs:OCRT_STRIP_TILES=$S -- terrain_2m_1080p_ao terrain_20m_1080p_ao bunny_1080p_ao interior_1080p_ao --reps 1 --frames 40 2>&1
done | tee $OUT/strip_sweep.log | cut -c1-175
