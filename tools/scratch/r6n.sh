cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for h in 1 0; do
echo "LDS history $h"
BU_RDO_LDS_HISTORY=$h timeout 300 python tools/rdo_lanes.py 12 2,3,4 0 2>&1 | grep lanes
BU_RDO_LDS_HISTORY=$h timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pipelined --no-big --no-uastc > gpurun_out/r6n_h$h.json 2>gpurun_out/r6n.err
python - <<P
import json
d=json.loads(open('gpurun_out/r6n_h$h.json').read().strip().splitlines()[-1])
u=d['uastc_rdo']; print(u['value'], u['ms_per_step'], u['images_identical_to_reference'], u['one_batch_start_to_finish'], u['serial_step_us'], u['kernels_ms_per_step'])
P
done
