mkdir -p gpurun_out/r05d
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05d/gputests.log 2>&1; tail -5 gpurun_out/r05d/gputests.log
timeout -k 10 200 python -m pytest tests/test_gpu_margin.py -s -q > gpurun_out/r05d/margin.txt 2>&1; python tools/margin_json.py gpurun_out/r05d/margin.txt gpurun_out/r05d/margin_families.json "round-5 kernels: parity = fp16 pair"
timeout -k 10 400 python bench.py > gpurun_out/r05d/bench_default.json 2> gpurun_out/r05d/bench_default.err; tail -3 gpurun_out/r05d/bench_default.err; python -c "
import json; d=json.load(open('gpurun_out/r05d/bench_default.json')); print(d['value'], d['api'], d['parity'])"
