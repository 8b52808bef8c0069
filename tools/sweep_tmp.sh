for W in bunny_1080p_ao bunny_600_defaults bunny_1080p_s16 interior_1080p_ao; do
  echo -n "base: "; timeout -k 5 100 python3 tools/prof_run.py --workload $W --frames 10 | tail -1
  echo -n "wg:   "; OCRT_LIB_DIR=lib_stamps timeout -k 5 100 python3 tools/prof_run.py --workload $W --frames 10 | tail -1
done
