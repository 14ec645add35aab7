mkdir -p gpurun_out/r05x
for sk in 0 4096 69632 1052672 0 4096 69632 1052672; do echo "== WSI_WS_SKEW $sk"; WSI_WS_SKEW=$sk timeout -k 10 100 python tools/launch_times.py --planes 3 --n 2000 2>&1 | tail -1; done > gpurun_out/r05x/skew.txt 2>&1
cat gpurun_out/r05x/skew.txt
