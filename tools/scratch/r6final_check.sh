cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
s=$(date +%s); timeout 900 python bench.py > gpurun_out/bench_check.json 2> gpurun_out/bench_check.err; echo "rc $? seconds $(( $(date +%s) - s ))"
python - <<P
import json
d=json.loads(open('gpurun_out/bench_check.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("metric","value","unit","n_gpus","steps","warmup","ms_per_step","identical_to_reference","host_gap_ms")}, d["roofline"]["frac"], d["cpu_baseline"]["value"])
print({k:(d.get(k) or {}).get("value") for k in ("uastc","uastc_rdo","etc1s_8192_q255","reference_default_threads","fast_codebooks","pipelined")})
P
