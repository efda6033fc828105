# One box, several settings of an environment knob: ETC1S step (and the 8192^2 leg with `big`).   gpurun -- bash tools/scratch/ab_env.sh VAR "v1 v2 ..." [big]
cd $GRAFT_REPO_ROOT
VAR=$1; VALS=$2; BIG=$3
for r in 1 2; do for v in $VALS; do
  echo -n "$VAR=$v: "
  env $VAR=$v timeout 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pipelined $([ -z "$BIG" ] && echo --no-big) --no-uastc --no-fast 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['kernels_ms_per_step']
print(d['value'], d['ms_per_step'], 'gap', d['host_gap_ms'], {x: k[x] for x in k if 'tsvq' in x}, (d.get('etc1s_8192_q255') or {}).get('value'))"
done; done
