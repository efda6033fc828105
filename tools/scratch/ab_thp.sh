cd $GRAFT_REPO_ROOT
for r in 1 2; do for t in glibc.malloc.hugetlb=0 glibc.malloc.hugetlb=1; do
  echo -n "$t: "
  GLIBC_TUNABLES=$t timeout 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-big --no-uastc --no-fast 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], 'gap', d['host_gap_ms'], 'pipelined', d['pipelined']['value'], 'with backend', d['pipelined_with_backend']['value'], 'backend ms', d['backend']['ms_per_image'])"
done; done
