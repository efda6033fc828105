# The rounds of both tree builds of one frontend step (BU_TSVQ_ROUNDS=1: one stderr line per device round) and the kernel time line of that step.
# usage (on the GPU box): tools/scratch/tsvq_rounds.sh <tag> [codebook threads]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
tag=${1:-x}; thr=${2:-0}
cat > /tmp/one_step.py <<PY
import sys, os, numpy as np
sys.path.insert(0, "$GRAFT_REPO_ROOT"); sys.path.insert(0, "$GRAFT_REPO_ROOT/tests")
import helpers
from basis_universal_amd import capi
from basis_universal_amd.etc1s import Etc1sFrontend, quality_to_clusters
size = int(os.environ.get("SIZE", "4096")); q = int(os.environ.get("QUALITY", "128"))
blocks = helpers.to_pixel_blocks(helpers.synth(size, size, 1234 if size == 4096 else 5678))
ctx = capi.Context(0)
ep, sel = quality_to_clusters(q, blocks.shape[0])
for i in range(int(os.environ.get("STEPS", "3"))):
    if i == int(os.environ.get("STEPS", "3")) - 1: os.environ["BU_TSVQ_ROUNDS"] = "1"
    fe = Etc1sFrontend(ctx, max_threads=$thr); fe.init(blocks, ep, sel, 1, True); fe.compress()
    if i == int(os.environ.get("STEPS", "3")) - 1: print({k: round(v * 1e3, 2) for k, v in fe.stage_times()})
    fe.close()
ctx.close()
PY
python /tmp/one_step.py > gpurun_out/rounds_$tag.txt 2>&1
cd /tmp && export TMPDIR=/tmp
STEPS=4 timeout 120 rocprofv3 --kernel-trace -d /tmp/tl_$tag -o tl -- python /tmp/one_step.py > /dev/null 2>&1
db=$(ls /tmp/tl_$tag/*/*.db /tmp/tl_$tag/*.db 2>/dev/null | head -1)
python $GRAFT_REPO_ROOT/tools/rocprof_timeline.py $db > $GRAFT_REPO_ROOT/gpurun_out/timeline_$tag.txt 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/timeline_$tag.txt
