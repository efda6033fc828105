cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_frontend_pipeline.py -x -q 2>&1 | tail -2
for a in "--streams 4" "--streams 4 --drivers 2" "--streams 6" "--streams 6 --drivers 2" "--streams 6 --drivers 3" "--streams 8 --drivers 2"; do timeout 200 python tools/inflight_probe.py --pipeline $a 2>&1 | grep in_flight | cut -c1-330; done
timeout 200 python tools/inflight_probe.py --streams 4,6 2>&1 | grep in_flight | cut -c1-250
