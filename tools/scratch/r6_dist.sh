cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for w in 32768 0; do
BU_CODEBOOK_WIDE_MIN=$w timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pipelined --no-uastc --no-fast > gpurun_out/r6h_w$w.json 2> gpurun_out/r6h_w$w.err
python - <<P
import json
d=json.loads(open('gpurun_out/r6h_w$w.json').read().strip().splitlines()[-1])
print('wide_min $w | headline', d['value'], d['ms_per_step'], d['identical_to_reference'], d['kernels_ms_per_step'].get('generate_endpoint_codebook'))
for k in ('etc1s_8192_q255','etc1s_noise4096_q128','etc1s_kodak4096_q128','etc1s_cube4096_q128'):
    b=d[k]; print('   ', k, b['value'], b['ms_per_step'], b['identical_to_reference'], 'codebook', b['kernels_ms_per_step'].get('generate_endpoint_codebook'))
P
done 2>&1 | tee gpurun_out/r6h_dist.log
