mkdir -p gpurun_out/r05e
timeout -k 10 600 python -m pytest tests/test_gpu_trunk.py tests/test_gpu_kernels.py -x -q > gpurun_out/r05e/tests.log 2>&1; tail -5 gpurun_out/r05e/tests.log
grep -q "passed" gpurun_out/r05e/tests.log && ! grep -q "failed" gpurun_out/r05e/tests.log || exit 1
for m in 1 131073 1 131073; do echo "== wsi_conv_set_mode $m"; timeout -k 10 200 python tools/launch_times.py --planes 3 --n 2000 --s2 $m | tail -9; done > gpurun_out/r05e/launch_times_d8_ab.txt 2>&1
cat gpurun_out/r05e/launch_times_d8_ab.txt
timeout -k 10 600 bash tools/pmc_trunk.sh r05e_trunk --planes 3 --n 2000 && cp gpurun_out/pmc_r05e_trunk/summary.txt gpurun_out/r05e/trunk_kernels_counters.txt
grep -A12 "wide_kernel" gpurun_out/r05e/trunk_kernels_counters.txt | grep "grid\|conflict\|matrix pipe"
timeout -k 10 400 python bench.py > gpurun_out/r05e/bench_default.json 2> gpurun_out/r05e/bench_default.err; python -c "
import json; d=json.load(open('gpurun_out/r05e/bench_default.json')); print(d['value'], d['api'], d['roofline']['achieved'], {k: round(v['avg_ms'],3) for k,v in d['kernels'].items()})"
