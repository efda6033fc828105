# deep rounds: parity of the codebook builders and the frontend, then the headline step with 0 / 1 / 2 generations of descendants per round
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -X faulthandler -m pytest tests/test_gpu_tsvq.py -x -q > gpurun_out/r6c_tsvq.log 2>&1; tail -5 gpurun_out/r6c_tsvq.log
timeout 900 python -m pytest tests/test_gpu_etc1s_frontend.py tests/test_gpu_baseline_configs.py -x -q -k "not uastc" 2>&1 | tail -5 > gpurun_out/r6c_tests.log
for d in 2 0 2 0; do
  BU_TSVQ_DEEP=$d timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pipelined --no-big > gpurun_out/r6c_bench_deep$d.json 2> gpurun_out/r6c_bench_deep$d.err
  python - <<P
import json
d=json.loads(open('gpurun_out/r6c_bench_deep$d.json').read().strip().splitlines()[-1])
print('deep $d', d['value'], d['ms_per_step'], d.get('identical_to_reference'), d.get('host_gap_ms'), {k:v for k,v in d.get('kernels_ms_per_step',{}).items() if 'tsvq' in k})
P
done 2>&1 | tee gpurun_out/r6c_ab.log
cat gpurun_out/r6c_tests.log
