W="bunny_1080p_s16 bunny_1080p_s64 interior_4k_ao bunny_1080p_s4"
run() { for w in $W; do echo -n "$1: "; env $2 python3 tools/prof_run.py --workload $w --frames 8 | tail -1; done; }
run rule "X=1"
run claim7 "OCRT_AO_CLAIM_MAX=7"
run claim14 "OCRT_AO_CLAIM_MAX=14"
run claim28 "OCRT_AO_CLAIM_MAX=28"
