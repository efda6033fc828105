# Host backend with 1 / 2 / 3 / 8 host threads on the GPU box's host: with one thread the walks run one after the other, so each loop's OWN cost shows
# (that is how the three-thread pipeline of rounds 1-2 was found to cost more than it hid).   gpurun -- bash tools/scratch/ab_backend_threads.sh
cd $GRAFT_REPO_ROOT
for t in 1 2 3 8; do
  echo -n "BU_HOST_THREADS=$t: "
  BU_HOST_THREADS=$t timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pipelined --no-big --no-uastc --no-fast 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); b = d['backend']
print(b['ms_per_image'], {k: round(v * 1000, 1) for k, v in b['stages_s'].items() if v > 0.002})"
done
