cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
BU_CB_TIMES=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pipelined --no-uastc --no-fast --no-big > gpurun_out/r6cb.json 2>gpurun_out/r6cb.err
grep "cb times" gpurun_out/r6cb.err | tail -32
