mkdir -p gpurun_out/r05ao
timeout -k 10 600 python -m pytest tests/test_gpu_unet.py -m gpu -x -q -s > gpurun_out/r05ao/gputests.log 2>&1; grep -E "x0 from|passed|failed|Error|error|assert" gpurun_out/r05ao/gputests.log | head -20
O=$GRAFT_REPO_ROOT/gpurun_out/r05ao
cd /tmp && export TMPDIR=/tmp
for f in 0 8388608; do
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace$f -o seg -- python3 $GRAFT_REPO_ROOT/tools/seg_once.py --reps 4 --set-mode $f > $O/trace$f.log 2>&1
python3 - $O/trace$f $f <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
st = [i for i, n in enumerate(names) if 'stem_pool' in n]
en = [i for i, n in enumerate(names) if 'unet_tail' in n]
tot = 0
for r in rows[st[-1]:en[-1] + 1]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    if 'stem' in r['Kernel_Name']:
        print(sys.argv[2], r['Kernel_Name'][:60], r['Grid_Size_X'], r['Workgroup_Size_X'], '%.1f us' % d)
print(sys.argv[2], 'batch sum %.1f us over %d launches' % (tot, en[-1] + 1 - st[-1]))
PY
done
