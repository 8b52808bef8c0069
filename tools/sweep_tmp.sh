for W in bunny_1080p_ao bunny_600_defaults interior_1080p_ao interior_4k_ao bunny_1080p_s4 bunny_1080p_s16 bunny_1080p_s64; do
  echo -n "$W: "; OCRT_PRINT_COST=1 python3 tools/prof_run.py --workload $W --frames 1 2>&1 | grep "mean cost"
done
