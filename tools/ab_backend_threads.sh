cd $GRAFT_REPO_ROOT
for t in 1 2 3 8; do
  echo -n "BU_HOST_THREADS=$t: "
  BU_HOST_THREADS=$t timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pipelined --no-big --no-uastc --no-fast 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); b = d['backend']
print(b['ms_per_image'], {k: round(v * 1000, 1) for k, v in b['stages_s'].items() if v > 0.002})"
done
