cd $GRAFT_REPO_ROOT
timeout -k 10 200 python tools/launch_times.py --planes 3 | tr '\n' ' ' | sed 's/TFLOP\/s//g; s/conv3x3_//g; s/   */ /g'; echo
timeout -k 10 200 python tools/launch_times.py --planes 3 --s2 33 | tr '\n' ' ' | sed 's/TFLOP\/s//g; s/conv3x3_//g; s/   */ /g'; echo
timeout -k 10 500 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "stride1" 2>&1 | tail -2
