for C in 0 2 4; do
  echo "claim_max=$C (0 = rule):"; env $( [ $C = 0 ] && echo X=1 || echo OCRT_AO_CLAIM_MAX=$C ) python3 tools/partition_probe.py bunny_1080p_ao 2,4,8 2>&1 | tail -3
done
for C in 0 2 4; do
  echo "interior claim_max=$C:"; env $( [ $C = 0 ] && echo X=1 || echo OCRT_AO_CLAIM_MAX=$C ) python3 tools/partition_probe.py interior_1080p_ao 4,8 2>&1 | tail -2
done
