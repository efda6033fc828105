import sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, helpers
from basis_universal_amd import capi
from basis_universal_amd.etc1s import Etc1sFrontend, quality_to_clusters
ctx = capi.Context(0)
blocks = helpers.to_pixel_blocks(helpers.synth(4096, 4096, 1234))
ep, sel = quality_to_clusters(128, blocks.shape[0])
fe = Etc1sFrontend(ctx); fe.init(blocks, ep, sel, 1, True); fe.compress()
idx = fe.get("block_endpoint_clusters_indices", np.uint32)
sizes = np.bincount(idx) * 16
print("clusters", sizes.size, "texels: min", sizes.min(), "median", int(np.median(sizes)), "mean", int(sizes.mean()), "max", sizes.max())
for t in (512, 1024, 2048, 4096, 8192, 16384, 32768):
    m = sizes < t
    print(f"< {t:6d} texels: {m.sum():5d} clusters, {100.0 * sizes[m].sum() / sizes.sum():5.1f} % of the texels")
