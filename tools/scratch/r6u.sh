cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_reference_seam.py -m gpu -x -q 2>&1 | tail -3
python - <<P
import sys; sys.path.insert(0,'tests')
import numpy as np, helpers
helpers.synth(4096,4096,1234).tofile('/tmp/img.rgba')
P
for t in 1 8; do oracle/_ref/process_bench_resident /tmp/img.rgba 4096 4096 128 1 $t 1 6 | tail -1; done
