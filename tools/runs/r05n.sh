mkdir -p gpurun_out/r05n
timeout -k 10 600 bash tools/pmc_trunk.sh r05n_speed --planes 1 --n 2000 && cp gpurun_out/pmc_r05n_speed/summary.txt gpurun_out/r05n/trunk_kernels_counters_speed.txt
grep -B1 -A28 "wide_kernel<1" gpurun_out/r05n/trunk_kernels_counters_speed.txt | grep "grid\|->\|INSTS_MFMA\|INSTS_VALU\|WAIT_ANY\|WAVE_CYCLES\|INSTS_LDS"
timeout -k 10 100 python tools/launch_times.py --planes 1 --n 2000 > gpurun_out/r05n/launch_times_speed.txt 2>&1; tail -19 gpurun_out/r05n/launch_times_speed.txt
