mkdir -p gpurun_out/r04p
timeout -k 10 600 python -m pytest tests/test_gpu_proposals.py tests/test_gpu_boundary.py -x -q > gpurun_out/r04p/tests.log 2>&1; echo tests rc=$?; tail -5 gpurun_out/r04p/tests.log
