mkdir -p gpurun_out/r05a
tools/probes/f16_denorm > gpurun_out/r05a/f16_denorm.txt 2>&1
cat gpurun_out/r05a/f16_denorm.txt
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r05a/gputests.log 2>&1; tail -3 gpurun_out/r05a/gputests.log
timeout -k 10 300 python bench.py > gpurun_out/r05a/bench_default.json 2> gpurun_out/r05a/bench_default.err; cut -c1-400 gpurun_out/r05a/bench_default.json
