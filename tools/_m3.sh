cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_trunk.py -x -q -m gpu -k "edge" 2>&1 | tail -5
