mkdir -p gpurun_out/r05l
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05l/gputests.log 2>&1; tail -4 gpurun_out/r05l/gputests.log
grep -q "passed" gpurun_out/r05l/gputests.log && ! grep -q "failed\|error" gpurun_out/r05l/gputests.log || exit 1
bash tools/collect_r05.sh r05l > gpurun_out/r05l/collect.log 2>&1; tail -30 gpurun_out/r05l/collect.log
