mkdir -p gpurun_out/r05ac
timeout -k 10 500 python bench.py --workload seg --no-cpu-baseline > gpurun_out/r05ac/bench_seg.json 2> gpurun_out/r05ac/bench_seg.err; tail -3 gpurun_out/r05ac/bench_seg.err
python -c "
import json; d=json.load(open('gpurun_out/r05ac/bench_seg.json')); a=d['api']; print(d['value'], a['value'], a['ms_per_slide'], a['vs_bare_engine_one_slide_per_call'], a['generic_iterator_path'])"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05ac/trace -o seg -- python3 $GRAFT_REPO_ROOT/bench.py --workload seg --no-cpu-baseline --no-api-leg --no-prof --steps 2 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/r05ac/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r05ac/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last 64 launches = the last batch of the last step
out = open('gpurun_out/r05ac/seg_last_batch.txt', 'w')
for r in rows[-64:]:
    out.write('%-70s grid %s wg %s lds %s  %.1f us\n' % (r['Kernel_Name'][:70], r.get('Grid_Size_X', r.get('Grid_Size')), r.get('Workgroup_Size_X', r.get('Workgroup_Size')), r.get('LDS_Block_Size', ''), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
out.close()
import os
for p in glob.glob('gpurun_out/r05ac/trace/**/*', recursive=True):
    if os.path.isfile(p) and os.path.getsize(p) > 4e6: os.remove(p)
PY
cat gpurun_out/r05ac/seg_last_batch.txt
