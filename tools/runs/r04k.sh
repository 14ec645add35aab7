mkdir -p gpurun_out/r04k
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04k/gputests.log 2>&1; echo tests rc=$?; tail -4 gpurun_out/r04k/gputests.log
