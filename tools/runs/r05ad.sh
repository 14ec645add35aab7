mkdir -p gpurun_out/r05ad
timeout -k 10 600 python -m pytest tests/test_gpu_unet.py -m gpu -x -q -s > gpurun_out/r05ad/gputests.log 2>&1; grep -E "fused tail|passed|failed|Error|error" gpurun_out/r05ad/gputests.log | head -20
timeout -k 10 500 python bench.py --workload seg --no-cpu-baseline > gpurun_out/r05ad/bench_seg.json 2> gpurun_out/r05ad/bench_seg.err; tail -3 gpurun_out/r05ad/bench_seg.err
python -c "
import json; d=json.load(open('gpurun_out/r05ad/bench_seg.json')); a=d['api']; print(d['value'], d['ms_per_step'], d['contract'], a['value'], a['ms_per_slide']); print({k: (round(v['avg_ms'],3), v['launches']) for k,v in d['kernels'].items()})"
