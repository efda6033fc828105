cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q -k "uastc or ldr_table or smoke" 2>&1 | tail -3
for i in 1 2; do
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pipelined --no-fast --no-big > gpurun_out/r6u.json 2>gpurun_out/r6u.err
python - <<P
import json
d=json.loads(open('gpurun_out/r6u.json').read().strip().splitlines()[-1])
u=d['uastc']; r=d['uastc_rdo']
print(u['value'], u['ms_per_step'], u['identical_to_reference'], u['kernels_ms_per_step'], '| rdo', r['value'], r['images_identical_to_reference'], r['kernels_ms_per_step'])
P
done
