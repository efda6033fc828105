cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_etc1s_frontend.py tests/test_gpu_baseline_configs.py tests/test_gpu_etc1s_kernels.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do
timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-pipelined --no-uastc --no-fast > gpurun_out/r6z.json 2>gpurun_out/r6z.err
python - <<P
import json
d=json.loads(open('gpurun_out/r6z.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['identical_to_reference'], d['host_gap_ms'], d['single_launch_kernels_ms_in_timed_steps']['refine_endpoint_clusterization'])
for k in ('etc1s_8192_q255','etc1s_noise4096_q128','etc1s_kodak4096_q128','etc1s_cube4096_q128'):
    o=d.get(k) or {}; print(k, o.get('value'), o.get('identical_to_reference'), (o.get('kernels_ms_per_step') or {}).get('refine_endpoint_clusterization'), (o.get('kernels_ms_per_step') or {}).get('encode_etc1s_blocks'), (o.get('kernels_ms_per_step') or {}).get('generate_endpoint_codebook'))
P
done
