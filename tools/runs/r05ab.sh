mkdir -p gpurun_out/r05ab
timeout -k 10 600 python -m pytest tests/test_gpu_unet.py tests/test_gpu_multirank.py -m gpu -x -q > gpurun_out/r05ab/gputests.log 2>&1; tail -3 gpurun_out/r05ab/gputests.log
timeout -k 10 500 python bench.py --workload seg --no-cpu-baseline > gpurun_out/r05ab/bench_seg.json 2> gpurun_out/r05ab/bench_seg.err; tail -3 gpurun_out/r05ab/bench_seg.err
python -c "
import json; d=json.load(open('gpurun_out/r05ab/bench_seg.json')); print(d['value'], json.dumps(d['api'], indent=1))"
