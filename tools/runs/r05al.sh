mkdir -p gpurun_out/r05al
timeout -k 10 600 python -m pytest tests/test_gpu_unet.py -m gpu -x -q -s > gpurun_out/r05al/gputests.log 2>&1; grep -E "fused tail|passed|failed|Error|error|assert" gpurun_out/r05al/gputests.log | head -20
timeout -k 10 300 python tools/tail_stamps.py 128 > gpurun_out/r05al/stamps.txt 2>&1; cat gpurun_out/r05al/stamps.txt | grep -v Warn
O=$GRAFT_REPO_ROOT/gpurun_out/r05al
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o seg -- python3 $GRAFT_REPO_ROOT/tools/seg_once.py --reps 3 > $O/trace.log 2>&1
cd $GRAFT_REPO_ROOT
grep -h "unet_tail" $(find $O/trace -name "*kernel_trace.csv") | python3 -c "
import sys,csv
for r in csv.reader(sys.stdin):
    nums=[int(x) for x in r if x.isdigit() and len(x)>12]
    print([x for x in r if 'tail' in x][0][:40], (max(nums)-min(nums))/1e3)
"
