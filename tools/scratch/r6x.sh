cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_etc1s_kernels.py tests/test_gpu_etc1s_frontend.py tests/test_gpu_baseline_configs.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pipelined --no-uastc --no-fast > gpurun_out/r6x.json 2>gpurun_out/r6x.err
python - <<P
import json
d=json.loads(open('gpurun_out/r6x.json').read().strip().splitlines()[-1])
k=d['kernels_ms_per_step']
print(d['value'], d['ms_per_step'], d['identical_to_reference'], d['host_gap_ms'], 'encode', k['encode_etc1s_blocks'], 'codebook', k['generate_endpoint_codebook'], 'refine', k['refine_endpoint_clusterization'], 'find', k['find_optimal_selector_clusters'])
for n in ('etc1s_8192_q255','etc1s_noise4096_q128','etc1s_kodak4096_q128','etc1s_cube4096_q128','reference_default_threads'):
    b=d.get(n) or {}; kk=b.get('kernels_ms_per_step') or {}; print(n, b.get('value'), b.get('ms_per_step'), b.get('identical_to_reference'), kk.get('encode_etc1s_blocks'), kk.get('generate_endpoint_codebook'), kk.get('refine_endpoint_clusterization'))
P
