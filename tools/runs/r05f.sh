mkdir -p gpurun_out/r05f
for m in 1 262145 524289 1 262145 524289; do echo "== wsi_conv_set_mode $m"; timeout -k 10 200 python tools/launch_times.py --planes 3 --n 2000 --s2 $m | tail -18; done > gpurun_out/r05f/launch_times_prio_ab.txt 2>&1
grep "==\|sum of" gpurun_out/r05f/launch_times_prio_ab.txt
