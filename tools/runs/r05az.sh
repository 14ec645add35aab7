O=$GRAFT_REPO_ROOT/gpurun_out/r05az; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 tools/seg_once.py --reps 1 --n 512 > $O/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 tools/seg_once.py --reps 1 --n 512 > $O/write.log 2>&1
python3 - <<'PY'
import csv, glob, collections
acc = collections.OrderedDict()
for which in ('fetch', 'write'):
    for f in glob.glob('gpurun_out/r05az/%s/**/*counter_collection.csv' % which, recursive=True):
        rows = list(csv.DictReader(open(f)))
        rows.sort(key=lambda r: int(r.get('Dispatch_Id', 0)))
        for r in rows:
            if r['Counter_Name'] not in ('FETCH_SIZE', 'WRITE_SIZE'):
                continue
            key = (int(r['Dispatch_Id']), r['Kernel_Name'].split('(')[0].replace('void ', '')[:64], r.get('Grid_Size', ''))
            acc.setdefault(key, {})[r['Counter_Name']] = float(r['Counter_Value'])
out = open('gpurun_out/r05az/traffic_seg.txt', 'w')
out.write('# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over tools/seg_once.py --reps 1 --n 512 (parity): HBM bytes per launch,\n# FETCH x2 (gfx950), KiB units; per tile = / 512\n')
for (did, name, grid), v in acc.items():
    if not any(k in name for k in ('conv3x3', 'stem_pool', 'unet_tail', 'stem_conv')):
        continue
    rd, wr = 2.0 * 1024.0 * v.get('FETCH_SIZE', 0.0), 1024.0 * v.get('WRITE_SIZE', 0.0)
    out.write('%-66s grid %-9s read %8.1f MB  write %8.1f MB   per tile %6.2f + %6.2f MB\n' % (name, grid, rd / 1e6, wr / 1e6, rd / 512e6, wr / 512e6))
out.close()
PY
cat gpurun_out/r05az/traffic_seg.txt
find $O -name "*counter_collection.csv" -delete
