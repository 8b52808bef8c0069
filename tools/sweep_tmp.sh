for W in bunny_1080p_ao bunny_600_defaults bunny_1080p_s16 interior_1080p_ao interior_4k_ao; do
  for L in lib lib_w2 lib_w8; do echo -n "$L: "; OCRT_LIB_DIR=$L timeout -k 5 100 python3 tools/prof_run.py --workload $W --frames 12 | tail -1; done
done
