cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pipelined --no-uastc --no-fast > gpurun_out/r6stats.json 2>gpurun_out/r6stats.err
grep "refine stats" gpurun_out/r6stats.err | sort | uniq -c | head -20
