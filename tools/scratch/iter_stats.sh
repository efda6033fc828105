cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
cat > /tmp/one.py <<PY
import sys, os, numpy as np
sys.path.insert(0, "$GRAFT_REPO_ROOT"); sys.path.insert(0, "$GRAFT_REPO_ROOT/tests")
import helpers
from basis_universal_amd import capi
from basis_universal_amd.etc1s import Etc1sFrontend, quality_to_clusters
size = int(os.environ.get("SIZE", "4096")); q = int(os.environ.get("QUALITY", "128"))
blocks = helpers.to_pixel_blocks(helpers.synth(size, size, 1234 if size == 4096 else 5678))
ctx = capi.Context(0)
ep, sel = quality_to_clusters(q, blocks.shape[0])
fe = Etc1sFrontend(ctx, max_threads=0); fe.init(blocks, ep, sel, 1, True); fe.compress(); fe.close()
ctx.close()
PY
BU_TSVQ_STATS=1 BU_TSVQ_ROUNDS=1 timeout 300 python /tmp/one.py 2>&1 | grep "tsvq round\|wide node" | sed 's/| last pass.*//' > gpurun_out/iter_stats_4096.txt
SIZE=8192 QUALITY=255 BU_TSVQ_STATS=1 BU_TSVQ_ROUNDS=1 timeout 300 python /tmp/one.py 2>&1 | grep "tsvq round\|wide node" | sed 's/| last pass.*//' > gpurun_out/iter_stats_8192.txt
wc -l gpurun_out/iter_stats_*.txt
