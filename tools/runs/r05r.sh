mkdir -p gpurun_out/r05r
timeout -k 10 600 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_multirank.py tests/test_gpu_unet.py -x -q > gpurun_out/r05r/tests.log 2>&1; tail -3 gpurun_out/r05r/tests.log
timeout -k 10 400 python bench.py --no-cpu-baseline --no-parity-leg --no-bf16-leg > gpurun_out/r05r/bench_default.json 2> gpurun_out/r05r/bench_default.err; python -c "
import json; d=json.load(open('gpurun_out/r05r/bench_default.json')); print(d['value'], d['api']['value'], d['api']['vs_headline'], d['api']['ms_per_slide'], d['ms_per_step'])"
