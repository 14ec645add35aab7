mkdir -p gpurun_out/r05h
timeout -k 10 120 python -m pytest tests/test_gpu_trunk.py -x -q -k "persistent_layer1 or 96_byte" > gpurun_out/r05h/tests.log 2>&1; tail -3 gpurun_out/r05h/tests.log
for m in 1 2097153 4194305 6291457 1 2097153 4194305 6291457; do echo "== wsi_conv_set_mode $m"; timeout -k 10 100 python tools/launch_times.py --planes 3 --n 2000 --s2 $m > /tmp/lt.txt 2>&1; sed -n 2,6p /tmp/lt.txt; tail -1 /tmp/lt.txt; done > gpurun_out/r05h/launch_times_l2pf_ab.txt 2>&1
cat gpurun_out/r05h/launch_times_l2pf_ab.txt
