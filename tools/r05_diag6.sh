# round 5, sixth GPU call: UASTC RDO walks (lean build for unflagged strips + the build with the refit for flagged ones, two streams): parity and the Kodak batch
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_uastc_rdo.py tests/test_gpu_kodak24.py -x -q --durations=3 2>&1 | tail -8 > gpurun_out/r05f_tests.txt; tail -2 gpurun_out/r05f_tests.txt
timeout 300 python -m pytest tests/test_gpu_ldr_table.py -x -q -k "uastc" 2>&1 | tail -1
timeout 500 python - <<'PY' > gpurun_out/r05f_rdo_bench.txt 2>&1
import sys, json, types
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch; torch.zeros(1).cuda()
import bench, helpers
from basis_universal_amd import capi
ctx = capi.Context(0)
args = types.SimpleNamespace(steps=10, warmup=2, no_cpu_baseline=True)
r = bench.uastc_rdo_bench(ctx, helpers, args)
print(json.dumps({k: r[k] for k in ("value", "ms_per_step", "one_batch_start_to_finish", "kernels_ms_per_step", "serial_step_us", "images_identical_to_reference", "lanes_identical")}))
PY
tail -1 gpurun_out/r05f_rdo_bench.txt | cut -c1-1200
