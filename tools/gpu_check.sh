#!/bin/bash
# One gpurun call's worth of checking (GPU box): the -m gpu suite, an A/B of library builds on four workloads, the bench line.
#   tools/gpu_check.sh TAG [lib_a lib_b ...]      -> gpurun_out/TAG_pytest.txt, TAG_ab.txt, TAG_bench.json
tag=${1:-check}; shift
variants=${@:-lib}
python3 -m pytest tests -m gpu -x -q > gpurun_out/${tag}_pytest.txt 2>&1; tail -5 gpurun_out/${tag}_pytest.txt
python3 tools/ab_variants.py $variants -- bunny_1080p_ao bunny_1080p_primary interior_1080p_ao bunny_600_defaults --reps 2 > gpurun_out/${tag}_ab.txt 2>&1; cat gpurun_out/${tag}_ab.txt
python3 bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; tail -c 1500 gpurun_out/${tag}_bench.err
python3 -c "
import json; d=json.load(open('gpurun_out/${tag}_bench.json')); print('value', d['value'], 'ms', d['ms_per_step'], 'pipelined', d['pipelined']['value'], d.get('end_to_end'))"
