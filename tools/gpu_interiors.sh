#!/bin/bash
# GPU box: the -m gpu suite, then the bench line of the four interior workloads (stand-in and harder stand-in, 1080p and 4K).
tag=${1:-interiors}
python3 -m pytest tests -m gpu -x -q > gpurun_out/${tag}_pytest.txt 2>&1; tail -4 gpurun_out/${tag}_pytest.txt
for w in interior_hard_1080p_ao interior_hard_4k_ao interior_1080p_ao interior_4k_ao; do
  python3 bench.py --steps 20 --warmup 5 --workload $w --no-end-to-end > gpurun_out/${tag}_bench_$w.json 2> gpurun_out/${tag}_bench_$w.err
  python3 -c "
import json; d=json.load(open('gpurun_out/${tag}_bench_$w.json')); print('$w', d['value'], d['ms_per_step'], 'pipelined', d['pipelined']['value'], 'cpu', d['cpu_baseline']['value'], d['config']['rays_per_frame'])"
done
