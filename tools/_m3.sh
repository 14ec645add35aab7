cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
for m in 1 129 1 129; do echo "mode $m"; timeout -k 10 200 python tools/launch_times.py --planes 3 --s2 $m | grep "s2\|sum" | tr '\n' ' '; echo; done
timeout -k 10 200 python tools/launch_times.py --planes 2 | tail -1
