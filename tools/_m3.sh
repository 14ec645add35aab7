set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_gpu_trunk.py -x -q -m gpu -s -k "u8_slide or golden or edge or unfused" 2>&1 | grep -v "^$" | tail -14
