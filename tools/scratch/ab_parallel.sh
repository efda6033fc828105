# basis_parallel_compress over the resident build (oracle/_ref/process_bench_resident), one box, several settings:  gpurun -- bash tools/scratch/ab_parallel.sh
cd $GRAFT_REPO_ROOT
python - <<PY
import sys; sys.path.insert(0, "tests"); import helpers
helpers.synth(4096, 4096, 1234).tofile("/tmp/img4096.raw")
PY
run() {  # poll mode, host threads, images in flight
  echo -n "poll=$1 host_threads=$2 images=$3: "
  BU_TSVQ_POLL=$1 BU_HOST_THREADS=$2 timeout 300 oracle/_ref/process_bench_resident /tmp/img4096.raw 4096 4096 128 1 $3 1 3 - $3 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); b = min(d['call_s']); print(round($3 * 16.777216 / b, 1), 'Mpix/s', [round(x, 3) for x in d['call_s']], d['all_images_identical'])"
}
for r in 1 2; do
  run spin 2 16; run yield 2 16; run yield 1 16; run yield 2 24; run yield 1 24; run yield 2 12; run yield 4 16
done
