cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_tsvq.py -m gpu -x -q 2>&1 | tail -4
