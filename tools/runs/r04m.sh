mkdir -p gpurun_out/r04m
timeout -k 10 600 python -m pytest tests/test_gpu_unet.py -x -q > gpurun_out/r04m/tests.log 2>&1; echo tests rc=$?; tail -3 gpurun_out/r04m/tests.log
for m in 1 65537 1; do python bench.py --workload seg --s2 $m --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('seg mode $m', d['value'], d['ms_per_step'], {k:(round(v['avg_ms'],3), round(v['share_of_step'],3)) for k,v in d['kernels'].items() if 'unet' in k}, 'mx leg', d['parity']['value'])"; done | tee gpurun_out/r04m/seg_ab.log
