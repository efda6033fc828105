cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_etc1s_frontend.py tests/test_gpu_baseline_configs.py tests/test_gpu_backend.py tests/test_gpu_frontend_pipeline.py tests/test_gpu_uastc.py tests/test_gpu_uastc_rdo.py -m gpu -x -q 2>&1 | tail -5
for f in 1 0; do
BU_UASTC_FUSED_SCORE=$f timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pipelined --no-big --no-fast > gpurun_out/r6m_f$f.json 2>gpurun_out/r6m.err
python - <<P
import json
d=json.loads(open('gpurun_out/r6m_f$f.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['identical_to_reference'], d['host_gap_ms'], d['h2d_inclusive']['pageable']['value'], d['h2d_inclusive']['pinned']['value'])
u=d['uastc']; print('fused $f', u['value'], u['ms_per_step'], u['identical_to_reference'], u['kernels_ms_per_step'])
u=d['uastc_rdo']; print(u['value'], u['ms_per_step'], u['images_identical_to_reference'])
P
done
