cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -15 > gpurun_out/r6i_tests.log; tail -3 gpurun_out/r6i_tests.log
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/r6i_bench.json 2> gpurun_out/r6i_bench.err; tail -c 300 gpurun_out/r6i_bench.err
