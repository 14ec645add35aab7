mkdir -p gpurun_out/r05aj
timeout -k 10 300 python tools/tail_stamps.py 128 > gpurun_out/r05aj/stamps.txt 2>&1; cat gpurun_out/r05aj/stamps.txt | grep -v Warn
