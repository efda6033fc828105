cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() {
timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-pipelined --no-uastc --no-fast --no-big > gpurun_out/r6z.json 2>gpurun_out/r6z.err
python - <<P
import json
d=json.loads(open('gpurun_out/r6z.json').read().strip().splitlines()[-1])
print("$1", d['value'], d['ms_per_step'], d['identical_to_reference'], d['host_gap_ms'], d['kernels_ms_per_step']['tsvq_split_packed16_wide'], d['kernels_ms_per_step']['tsvq_split_packed16'])
P
}
for i in 1 2; do
BU_TSVQ_WIDE_COV_MIN=98304 run cov98304
BU_TSVQ_WIDE_COV_MIN=65536 run cov65536
BU_TSVQ_WIDE_COV_MIN=49152 run cov49152
BU_TSVQ_WIDE_MIN=6144 run wide6144
BU_TSVQ_WIDE_MIN=12288 run wide12288
BU_TSVQ_DENSE_MIN=129 run dense129
BU_TSVQ_DENSE_MIN=513 run dense513
done
