set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_trunk.py -x -q -m gpu -k "roundtrip or stride or fused or taps or mode3 or avgpool" 2>&1 | tail -3
timeout -k 10 300 python tools/tune_conv.py --n 1000 --planes 3 --cfgs 30,31,21 --ablate 1,3 --rounds 3 --iters 3 2>&1 | grep -v "^$" | tail -40
timeout -k 10 200 python tools/launch_times.py --planes 3 --s2 2
