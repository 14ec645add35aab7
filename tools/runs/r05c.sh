mkdir -p gpurun_out/r05c
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05c/gputests.log 2>&1; tail -15 gpurun_out/r05c/gputests.log
