# full parity suite + smoke, the reference's own -clbench through the HIP seam (oracle/_ref/basisu_hip), then the default bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
tag=${1:-r6}
timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/${tag}_tests.log; cat gpurun_out/${tag}_tests.log
timeout 120 python __graft_entry__.py smoke 2>&1 | tail -2
if [ -x oracle/_ref/basisu_hip ]; then (cd /tmp && timeout 300 $GRAFT_REPO_ROOT/oracle/_ref/basisu_hip -clbench > $GRAFT_REPO_ROOT/gpurun_out/${tag}_clbench.txt 2>&1; tail -12 $GRAFT_REPO_ROOT/gpurun_out/${tag}_clbench.txt); fi
timeout 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; tail -c 400 gpurun_out/${tag}_bench.json
