# deep rounds at 8192^2 q255 (16128 selector leaves: the replay handles thousands of nodes per round there)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for d in 0 2 1 0 2; do
  BU_TSVQ_DEEP=$d timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pipelined --no-uastc --no-fast > gpurun_out/r6g_deep$d.json 2> gpurun_out/r6g_deep$d.err
  python - <<P
import json
d=json.loads(open('gpurun_out/r6g_deep$d.json').read().strip().splitlines()[-1])
b=d['etc1s_8192_q255']
print('deep $d: 8192 q255', b['value'], b['ms_per_step'], b['identical_to_reference'], 'gap', b['host_gap_ms'], {k:v for k,v in b['kernels_ms_per_step'].items() if 'tsvq_split' in k}, '| 4096', d['value'], '| noise', (d.get('etc1s_noise4096_q128') or {}).get('value'), (d.get('etc1s_noise4096_q128') or {}).get('identical_to_reference'), '| kodak', (d.get('etc1s_kodak4096_q128') or {}).get('value'), '| cube', (d.get('etc1s_cube4096_q128') or {}).get('value'), (d.get('etc1s_cube4096_q128') or {}).get('identical_to_reference'))
P
done 2>&1 | tee gpurun_out/r6g_deep_8192.log
