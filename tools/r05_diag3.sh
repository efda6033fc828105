# round 5, third GPU call: pipeline parity (whole file), the reference's tools over the resident path (now through the pipeline), host-CPU variants of the pipeline, a full bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_frontend_pipeline.py tests/test_gpu_reference_seam.py tests/test_gpu_backend.py -x -q --durations=5 2>&1 | tail -15 > gpurun_out/r05c_tests.txt; tail -4 gpurun_out/r05c_tests.txt
for v in "" "BU_HOST_THREADS=1" "BU_HOST_THREADS=2" "BU_PIPELINE_SPIN_US=0" "BU_PIPELINE_SPIN_US=0 BU_PIPELINE_SLEEP_US=10" "GPU_MAX_HW_QUEUES=8"; do
  env $v timeout 200 python tools/inflight_probe.py --pipeline --streams 4 2>&1 | grep in_flight | cut -c1-520
done > gpurun_out/r05c_pipe_variants.txt; cat gpurun_out/r05c_pipe_variants.txt
timeout 600 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_r05c.json 2> gpurun_out/bench_r05c.err; tail -c 600 gpurun_out/bench_r05c.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_r05c.json"))
print({k:d.get(k) for k in ("value","ms_per_step","identical_to_reference","host_gap_ms","host_cpu_s_per_step")})
print(json.dumps(d.get("pipelined"))[:1500])
print(json.dumps(d.get("pipelined_with_backend"))[:500])
e=d.get("end_to_end",{}); print({k:e.get(k) for k in ("resident","resident_1_thread","resident_parallel","resident_parallel_glibc_hugetlb")})
print({k:(d.get(k) or {}).get("value") for k in ("uastc","uastc_rdo","etc1s_8192_q255","reference_default_threads","fast_codebooks")})
PY
