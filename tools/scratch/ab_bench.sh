# A/B of the ETC1S step between builds of the libraries on ONE box (the pool's boxes differ by a few per cent):  gpurun -- bash tools/scratch/ab_bench.sh "A B ..." [rounds]
# with gpurun_ab/<variant>/*.so prepared beforehand (cp basis_universal_amd/lib/*.so gpurun_ab/A/ ...). Prints value / ms_per_step / the TSVQ and sort labels per run.
cd $GRAFT_REPO_ROOT
for r in $(seq 1 ${2:-2}); do for v in $1; do
  cp gpurun_ab/$v/*.so basis_universal_amd/lib/
  echo -n "$v: "
  timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pipelined --no-big --no-uastc --no-fast 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['kernels_ms_per_step']
print(d['value'], d['ms_per_step'], 'gap', d['host_gap_ms'], {x: k[x] for x in k if 'tsvq' in x or 'rank' in x})"
done; done
