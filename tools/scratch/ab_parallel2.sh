cd $GRAFT_REPO_ROOT
cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag; ldd --version | head -1; nproc
python - <<PY
import sys; sys.path.insert(0, "tests"); import helpers
helpers.synth(4096, 4096, 1234).tofile("/tmp/img4096.raw")
PY
run() {  # tunables, poll mode, host threads, images in flight
  echo -n "tun=$1 poll=$2 host_threads=$3 images=$4: "
  GLIBC_TUNABLES=$1 BU_TSVQ_POLL=$2 BU_HOST_THREADS=$3 timeout 300 oracle/_ref/process_bench_resident /tmp/img4096.raw 4096 4096 128 1 $4 1 5 - $4 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); b = min(d['call_s']); print(round($4 * 16.777216 / b, 1), 'Mpix/s', [round(x, 3) for x in d['call_s']], d['all_images_identical'])"
}
for r in 1 2; do
  run glibc.malloc.hugetlb=0 yield 1 16; run glibc.malloc.hugetlb=1 yield 1 16; run glibc.malloc.hugetlb=1 spin 1 16; run glibc.malloc.hugetlb=1 yield 2 16; run glibc.malloc.hugetlb=1 yield 1 24
done
echo single image:
for t in glibc.malloc.hugetlb=0 glibc.malloc.hugetlb=1; do GLIBC_TUNABLES=$t oracle/_ref/process_bench_resident /tmp/img4096.raw 4096 4096 128 1 8 1 5 2>/dev/null | tail -1 | cut -c1-300; done
