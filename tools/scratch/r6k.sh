cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
BU_TSVQ_STATS=1 BU_TSVQ_ROUNDS=1 timeout 120 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pipelined --no-big --no-uastc --no-fast > /dev/null 2> gpurun_out/r6k_stats.log
for w in 0 1; do
BU_TSVQ_WINDOWS=$w timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pipelined --no-big --no-uastc --no-fast > gpurun_out/r6k_win$w.json 2>/dev/null
python - <<P
import json
d=json.loads(open('gpurun_out/r6k_win$w.json').read().strip().splitlines()[-1])
print('windows $w', d['value'], d['ms_per_step'], d['kernels_ms_per_step']['tsvq_split_packed16_wide'])
P
done
