cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() {
timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-pipelined --no-uastc --no-fast --no-big > gpurun_out/r6z.json 2>gpurun_out/r6z.err
python - <<P
import json
d=json.loads(open('gpurun_out/r6z.json').read().strip().splitlines()[-1])
print("$1", d['value'], d['ms_per_step'], d['identical_to_reference'], d['host_gap_ms'])
P
}
for i in 1 2 3 4; do BU_MAIL_FETCH=1 run mail1; BU_MAIL_FETCH=0 run mail0; done
