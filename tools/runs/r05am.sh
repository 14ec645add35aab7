mkdir -p gpurun_out/r05am
timeout -k 10 600 python -m pytest tests/test_gpu_unet.py -m gpu -x -q > gpurun_out/r05am/gputests.log 2>&1; tail -2 gpurun_out/r05am/gputests.log
for f in 0 8388608; do timeout -k 10 300 python tools/tail_stamps.py 128 $f 2>&1 | grep -v Warn | grep -v amdgpu.ids; done > gpurun_out/r05am/stamps.txt; cat gpurun_out/r05am/stamps.txt
O=$GRAFT_REPO_ROOT/gpurun_out/r05am
cd /tmp && export TMPDIR=/tmp
for f in 0 8388608; do
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace$f -o seg -- python3 $GRAFT_REPO_ROOT/tools/seg_once.py --reps 4 --set-mode $f > $O/trace$f.log 2>&1
grep -h "unet_tail" $(find $O/trace$f -name "*kernel_trace.csv") | python3 -c "
import sys,csv
for r in csv.reader(sys.stdin):
    nums=[int(x) for x in r if x.isdigit() and len(x)>12]
    print($f, [x for x in r if 'tail' in x][0][:50], (max(nums)-min(nums))/1e3)
"
done
