mkdir -p gpurun_out/r05b
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05b/gputests.log 2>&1; tail -15 gpurun_out/r05b/gputests.log
timeout -k 10 300 python -m pytest tests/test_gpu_unet.py tests/test_gpu_margin.py -s -q > gpurun_out/r05b/unet_margin.txt 2>&1; grep -i "planes\|dense\|mx\|parity\|max" gpurun_out/r05b/unet_margin.txt | cut -c1-220 | tail -30
timeout -k 10 300 python bench.py --workload seg > gpurun_out/r05b/bench_seg.json 2> gpurun_out/r05b/bench_seg.err; python -c "
import json; d=json.load(open('gpurun_out/r05b/bench_seg.json')); print(d['value'], d['contract'], d['parity'])"
timeout -k 10 300 python bench.py --mode parity --no-bf16-leg > gpurun_out/r05b/bench_parity.json 2> gpurun_out/r05b/bench_parity.err; python -c "
import json; d=json.load(open('gpurun_out/r05b/bench_parity.json')); print(d['value'], d['contract'], d['roofline'])"
