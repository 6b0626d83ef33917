cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_net.py -x -q -k smallest_batches 2>&1 | tail -30
