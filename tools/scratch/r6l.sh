cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_etc1s_kernels.py -m gpu -x -q -k "upload_and_encode" 2>&1 | tail -30
timeout 900 python -m pytest tests/test_gpu_etc1s_frontend.py tests/test_gpu_baseline_configs.py tests/test_gpu_backend.py tests/test_gpu_frontend_pipeline.py tests/test_gpu_etc1s_sharded.py -m gpu -x -q 2>&1 | tail -5
