# Host backend, two builds of the libraries side by side on ONE box's host (gpurun_ab/A, gpurun_ab/B prepared beforehand): ms per image and the sub-stage timers.
#   gpurun -- bash tools/scratch/ab_backend.sh
cd $GRAFT_REPO_ROOT
for r in 1 2; do for v in A B; do
  cp gpurun_ab/$v/*.so basis_universal_amd/lib/
  echo -n "$v: "
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pipelined --no-big --no-uastc --no-fast 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); b = d['backend']
print(b['ms_per_image'], b.get('identical_to_reference'), {k: round(v * 1000, 1) for k, v in b['stages_s'].items() if v > 0.002})"
done; done
