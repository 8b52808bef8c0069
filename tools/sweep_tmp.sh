for T in 0.3 0.4 0.5 0.6 0.7 0.85 2.0; do
  for W in bunny_1080p_ao bunny_1080p_s16 interior_1080p_ao; do
    echo -n "contract $T: "; OCRT_CONTRACT=$T python3 tools/prof_run.py --workload $W --frames 6 | tail -1
  done
done
