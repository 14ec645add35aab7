O=$GRAFT_REPO_ROOT/gpurun_out/r05as; mkdir -p $O
timeout -k 10 500 python bench.py --workload seg --no-cpu-baseline > $O/bench_seg.json 2> $O/bench_seg.err || tail -5 $O/bench_seg.err
python -c "
import json; d=json.load(open('$O/bench_seg.json')); a=d['api']; print(d['value'], d['ms_per_step'], a['value'], a['ms_per_slide'], a['vs_bare_engine_one_slide_per_call'], a['bare_pipeline_one_slide_per_call']['value'], a['vs_bare_pipeline_one_slide_per_call'])"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o seg -- python3 $GRAFT_REPO_ROOT/tools/seg_once.py --reps 3 --n 512 > $O/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r05as/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
st = [i for i, n in enumerate(names) if 'stem_pool' in n]
en = [i for i, n in enumerate(names) if 'unet_tail' in n]
out = open('gpurun_out/r05as/seg_batch512.txt', 'w')
out.write('# rocprofv3 --kernel-trace of tools/seg_once.py --n 512 (parity, one batch of 512 tiles of 256x256): the launches of ONE batch in issue order\n')
tot = 0
for r in rows[st[-1]:en[-1] + 1]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    out.write('%-72s grid %-9s wg %-4s %8.1f us\n' % (r['Kernel_Name'][:72], r['Grid_Size_X'], r['Workgroup_Size_X'], d))
out.write('sum %.1f us\n' % tot)
out.close()
PY
cat gpurun_out/r05as/seg_batch512.txt
find $O/trace -name "*.csv" -size +3M -delete
