cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for w in 3 10 3 20; do
timeout 300 python bench.py --steps 20 --warmup $w --no-cpu-baseline --no-pipelined --no-uastc --no-fast --no-big > gpurun_out/r6p.json 2>gpurun_out/r6p.err
python - <<P
import json
d=json.loads(open('gpurun_out/r6p.json').read().strip().splitlines()[-1])
print('warmup $w', d['value'], d['ms_per_step'], d['host_gap_ms'], d['instrumented_pass']['ms_per_step'], d['h2d_inclusive']['pageable']['value'], d['h2d_inclusive']['pinned']['value'])
P
done
