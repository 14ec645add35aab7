mkdir -p gpurun_out/r04o
timeout -k 10 800 python -m pytest tests/test_gpu_multirank.py -x -q > gpurun_out/r04o/tests.log 2>&1; echo tests rc=$?; tail -5 gpurun_out/r04o/tests.log
