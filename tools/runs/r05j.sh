mkdir -p gpurun_out/r05j
timeout -k 10 600 python -m pytest tests/test_gpu_trunk.py tests/test_gpu_margin.py tests/test_gpu_fullsize.py -x -q > gpurun_out/r05j/tests.log 2>&1; tail -5 gpurun_out/r05j/tests.log
grep -q "passed" gpurun_out/r05j/tests.log && ! grep -q "failed\|error" gpurun_out/r05j/tests.log || exit 1
for m in 1 16385 1048577 1; do echo "== wsi_conv_set_mode $m"; timeout -k 10 100 python tools/launch_times.py --planes 3 --n 2000 --s2 $m > /tmp/lt.txt 2>&1; sed -n 2,7p /tmp/lt.txt; tail -1 /tmp/lt.txt; done > gpurun_out/r05j/launch_times_planar96.txt 2>&1
cat gpurun_out/r05j/launch_times_planar96.txt
timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/r05j/bench_default.json 2> gpurun_out/r05j/bench_default.err; python -c "
import json; d=json.load(open('gpurun_out/r05j/bench_default.json')); print(d['value'], d['api']['value'], d['roofline']['achieved'], d['roofline_layer1']['achieved'], {k: round(v['avg_ms'],3) for k,v in d['kernels'].items()})"
