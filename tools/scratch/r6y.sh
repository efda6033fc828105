cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-pipelined --no-uastc --no-fast --no-big > gpurun_out/r6y.json 2>gpurun_out/r6y.err
python - <<P
import json
d=json.loads(open('gpurun_out/r6y.json').read().strip().splitlines()[-1])
print('$1', d['value'], d['ms_per_step'], d['identical_to_reference'], d['host_gap_ms'])
P
}
for i in 1 2 3; do
BU_TSVQ_DEEP=0 run "deep0"
BU_TSVQ_DEEP=1 BU_TSVQ_DEEP_MAX_NODES=64 run "deep1 max64"
BU_TSVQ_DEEP=1 BU_TSVQ_DEEP_MAX_NODES=32 run "deep1 max32"
BU_TSVQ_DEEP=1 BU_TSVQ_DEEP_MAX_NODES=100 run "deep1 max100"
done
