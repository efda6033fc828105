cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_etc1s_frontend.py tests/test_gpu_baseline_configs.py tests/test_gpu_backend.py tests/test_gpu_frontend_pipeline.py tests/test_gpu_etc1s_sharded.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2 3; do
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pipelined --no-uastc --no-fast --no-big > gpurun_out/r6o.json 2>gpurun_out/r6o.err
python - <<P
import json
d=json.loads(open('gpurun_out/r6o.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['identical_to_reference'], d['host_gap_ms'], d['instrumented_pass']['ms_per_step'], d['h2d_inclusive']['pageable']['value'], d['h2d_inclusive']['pinned']['value'])
P
done
BU_RESULT_PREFETCH=0 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pipelined --no-uastc --no-fast --no-big > gpurun_out/r6o.json 2>gpurun_out/r6o.err
python - <<P
import json
d=json.loads(open('gpurun_out/r6o.json').read().strip().splitlines()[-1])
print('no prefetch', d['value'], d['ms_per_step'], d['identical_to_reference'], d['host_gap_ms'])
P
