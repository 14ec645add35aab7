mkdir -p gpurun_out/r05ai
timeout -k 10 600 python -m pytest tests/test_gpu_unet.py -m gpu -x -q -s > gpurun_out/r05ai/gputests.log 2>&1; grep -E "fused tail|passed|failed|Error|error|assert" gpurun_out/r05ai/gputests.log | head -20
timeout -k 10 500 python bench.py --workload seg --no-cpu-baseline --no-api-leg > gpurun_out/r05ai/bench_seg.json 2> gpurun_out/r05ai/bench_seg.err; tail -3 gpurun_out/r05ai/bench_seg.err
python -c "
import json; d=json.load(open('gpurun_out/r05ai/bench_seg.json')); print(d['value'], d['ms_per_step']); print({k: (round(v['avg_ms'],3), v['launches']) for k,v in d['kernels'].items()})"
O=$GRAFT_REPO_ROOT/gpurun_out/r05ai
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o seg -- python3 $GRAFT_REPO_ROOT/tools/seg_once.py --reps 3 > $O/trace.log 2>&1
cd $GRAFT_REPO_ROOT
grep -h "unet_tail" $(find $O/trace -name "*kernel_trace.csv") | awk -F, '{print $0}' | python3 -c "
import sys,csv
for r in csv.reader(sys.stdin):
    nums=[int(x) for x in r if x.isdigit() and len(x)>12]
    print([x for x in r if 'tail' in x][0][:40], (max(nums)-min(nums))/1e3 if len(nums)>=2 else r)
"
