for rep in 1 2; do for W in bunny_1080p_ao bunny_600_defaults interior_1080p_ao interior_4k_ao; do for H in 2 3 4 5; do
python3 bench.py --steps 40 --warmup 10 --workload $W --in-flight $H --no-cpu-baseline --no-end-to-end --min-seconds 0.4 2>/dev/null | tail -1 | python3 -c "
import sys,json
b=json.loads(sys.stdin.read()); print('$W', $H, 'hosts:', b['value'], b['ms_per_step'], b['blocks']['min'], b['blocks']['max'])"
done; done; done
