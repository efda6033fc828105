cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_etc1s_kernels.py tests/test_gpu_etc1s_frontend.py tests/test_gpu_baseline_configs.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pipelined --no-uastc --no-fast > gpurun_out/r6x.json 2>gpurun_out/r6x.err
python - <<P
import json
d=json.loads(open('gpurun_out/r6x.json').read().strip().splitlines()[-1])
k=d['kernels_ms_per_step']; k8=d['etc1s_8192_q255']['kernels_ms_per_step']
print(d['value'], d['ms_per_step'], d['identical_to_reference'], 'codebook', k['generate_endpoint_codebook'], 'encode', k['encode_etc1s_blocks'], '8192', d['etc1s_8192_q255']['value'], k8['generate_endpoint_codebook'], 'noise', d['etc1s_noise4096_q128']['value'], d['etc1s_noise4096_q128']['kernels_ms_per_step']['generate_endpoint_codebook'], 'cube', d['etc1s_cube4096_q128']['value'])
P
done
