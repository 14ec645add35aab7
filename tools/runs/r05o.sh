mkdir -p gpurun_out/r05o
timeout -k 10 200 python -m pytest tests/test_gpu_trunk.py -x -q -k "next_line_prefetch or eight_pixel" > gpurun_out/r05o/tests.log 2>&1; tail -3 gpurun_out/r05o/tests.log
grep -q "passed" gpurun_out/r05o/tests.log && ! grep -q "failed\|error" gpurun_out/r05o/tests.log || exit 1
for m in 1 2097153 1 2097153; do echo "== wsi_conv_set_mode $m"; timeout -k 10 100 python tools/launch_times.py --planes 3 --n 2000 --s2 $m > /tmp/lt.txt 2>&1; sed -n 7,18p /tmp/lt.txt; tail -1 /tmp/lt.txt; done > gpurun_out/r05o/launch_times_widepf_ab.txt 2>&1
cat gpurun_out/r05o/launch_times_widepf_ab.txt
