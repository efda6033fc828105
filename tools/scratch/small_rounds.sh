cd $GRAFT_REPO_ROOT
cat > /tmp/one.py <<PY
import sys, os, numpy as np
sys.path.insert(0, "$GRAFT_REPO_ROOT"); sys.path.insert(0, "$GRAFT_REPO_ROOT/tests")
import helpers
from basis_universal_amd import capi
from basis_universal_amd.etc1s import Etc1sFrontend, quality_to_clusters
size = int(os.environ.get("SIZE", "1024"))
blocks = helpers.to_pixel_blocks(helpers.synth(size, size, 1234))
ctx = capi.Context(0)
ep, sel = quality_to_clusters(128, blocks.shape[0])
for i in range(3):
    if i == 2: os.environ["BU_TSVQ_ROUNDS"] = "1"
    fe = Etc1sFrontend(ctx, max_threads=0); fe.init(blocks, ep, sel, 1, True); fe.compress(); fe.close()
ctx.close()
PY
for s in 1024 1536; do echo size $s; SIZE=$s python /tmp/one.py 2>&1 | grep "tsvq round" | head -24 | cut -c1-110; done
