O=$GRAFT_REPO_ROOT/gpurun_out/r05bd; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_unet.py -m gpu -x -q > $O/gputests.log 2>&1; tail -2 $O/gputests.log
cd /tmp && export TMPDIR=/tmp
for f in 0 16777216; do
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace$f -o seg -- python3 $GRAFT_REPO_ROOT/tools/seg_once.py --reps 3 --n 512 --set-mode $f > $O/trace$f.log 2>&1
python3 - $O/trace$f $f <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
st = [i for i, n in enumerate(names) if 'stem_pool' in n]
en = [i for i, n in enumerate(names) if 'unet_tail' in n]
tot = 0
line = []
for r in rows[st[-1]:en[-1] + 1]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    if 'slab3_kernel<4, 1, 4' in r['Kernel_Name'] or ('slab3_kernel<4, 2, 2' in r['Kernel_Name'] and len(line) >= 2) or 'slab3_kernel<2, 8' in r['Kernel_Name']:
        line.append('%.0f' % d)
print(sys.argv[2], 'decoder slab3 launches (us):', ' '.join(line[-6:]), '| batch sum %.1f us' % tot)
PY
find $O/trace$f -name "*.csv" -size +2M -delete
done
