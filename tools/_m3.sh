set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -q -m gpu 2>&1 | tail -4
timeout -k 10 200 python tools/launch_times.py --planes 3 | tail -1
timeout -k 10 200 python tools/launch_times.py --planes 2 | tail -1
