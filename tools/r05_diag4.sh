# round 5, fourth GPU call: the whole parity suite (tuning struct, one tree per rank, pipeline), then the reference's drivers over the resident path with and without the pipeline
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -16 > gpurun_out/r05d_tests.txt; tail -3 gpurun_out/r05d_tests.txt
python - <<'PY'
import sys; sys.path.insert(0, "tests")
import numpy as np, helpers
np.ascontiguousarray(helpers.synth(4096, 4096, 1234)).tofile("/tmp/img4096.raw")
PY
for lanes in 0 4 6; do for thp in 0 1; do
  BU_RESIDENT_LANES=$lanes BU_HOST_THREADS=1 GLIBC_TUNABLES=glibc.malloc.hugetlb=$thp timeout 300 oracle/_ref/process_bench_resident /tmp/img4096.raw 4096 4096 128 1 16 1 4 - 16 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); b=min(d['call_s']); print('parallel16 lanes $lanes thp $thp:', round(16*16.777216/b,1), 'Mpix/s', [round(x,3) for x in d['call_s']], d['all_images_identical'], d['fnv1a64'])"
done; done > gpurun_out/r05d_resident_parallel.txt 2>&1; cat gpurun_out/r05d_resident_parallel.txt
for lanes in 0 4; do
  BU_RESIDENT_LANES=$lanes timeout 300 oracle/_ref/process_bench_resident /tmp/img4096.raw 4096 4096 128 1 1 1 4 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('single lanes $lanes:', [round(a+b,4) for a,b in zip(d['init_s'],d['process_s'])], d['fnv1a64'])"
done > gpurun_out/r05d_resident_single.txt 2>&1; cat gpurun_out/r05d_resident_single.txt
