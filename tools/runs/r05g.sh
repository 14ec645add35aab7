mkdir -p gpurun_out/r05g
timeout -k 10 120 python -m pytest tests/test_gpu_trunk.py -x -q -k "persistent_layer1 or 96_byte" > gpurun_out/r05g/tests_l1p.log 2>&1; tail -15 gpurun_out/r05g/tests_l1p.log
grep -q "passed" gpurun_out/r05g/tests_l1p.log && ! grep -q "failed\|error" gpurun_out/r05g/tests_l1p.log || exit 1
for m in 1 1048577 1 1048577; do echo "== wsi_conv_set_mode $m"; timeout -k 10 100 python tools/launch_times.py --planes 3 --n 2000 --s2 $m | head -8; done > gpurun_out/r05g/launch_times_l1p_ab.txt 2>&1
cat gpurun_out/r05g/launch_times_l1p_ab.txt
