# round 5, fifth GPU call: the UASTC RDO walk without the refit in it (lean walk + settle launches): parity, then timing of the phases and the pipelined batch
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_uastc_rdo.py tests/test_gpu_kodak24.py tests/test_gpu_uastc.py -x -q --durations=5 2>&1 | tail -12 > gpurun_out/r05e_tests.txt; tail -3 gpurun_out/r05e_tests.txt
timeout 300 python -m pytest tests/test_gpu_ldr_table.py -x -q -k "uastc" 2>&1 | tail -2
timeout 200 python tools/rdo_time.py 2048 4 1.0 > gpurun_out/r05e_rdo_time.txt 2>&1; tail -2 gpurun_out/r05e_rdo_time.txt | cut -c1-500
timeout 500 python - <<'PY' > gpurun_out/r05e_rdo_bench.txt 2>&1
import sys, json, types
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import bench, helpers
from basis_universal_amd import capi
ctx = capi.Context(0)
args = types.SimpleNamespace(steps=10, warmup=2, no_cpu_baseline=True)
print(json.dumps(bench.uastc_rdo_bench(ctx, helpers, args)))
PY
tail -1 gpurun_out/r05e_rdo_bench.txt | cut -c1-1500
