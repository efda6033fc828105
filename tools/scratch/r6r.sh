cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q --durations=6 2>&1 | tail -40 > gpurun_out/r6r_tests.log; grep -v "^\s*$" gpurun_out/r6r_tests.log | tail -14
