set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -q -m gpu 2>&1 | tail -25
