cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_tsvq.py tests/test_gpu_etc1s_frontend.py tests/test_gpu_baseline_configs.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pipelined --no-uastc --no-fast --no-big > gpurun_out/r6x.json 2>gpurun_out/r6x.err
python - <<P
import json
d=json.loads(open('gpurun_out/r6x.json').read().strip().splitlines()[-1])
k=d['kernels_ms_per_step']
print(d['value'], d['ms_per_step'], d['identical_to_reference'], d['host_gap_ms'], {x:k[x] for x in k if 'tsvq' in x})
P
done
