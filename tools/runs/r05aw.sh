O=$GRAFT_REPO_ROOT/gpurun_out/r05aw; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for d in _old .; do
tag=$(echo $d | tr -d './_'); tag=${tag:-new}
(cd $GRAFT_REPO_ROOT/$d && timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/tr_$tag -o c4 -- python3 bench.py --workload cfg4 --no-cpu-baseline --no-api-leg --no-prof --steps 3 --warmup 1 > $O/bench_$tag.json 2> $O/err_$tag.txt)
echo "== $tag"
python3 - $O/tr_$tag <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last step: from the last stem_pool launch on
st = [i for i, r in enumerate(rows) if 'stem_pool' in r['Kernel_Name']]
s = st[-1]
prev_end = None
for r in rows[s - 3:s + 40]:
    b, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (b - prev_end) / 1e3 if prev_end else 0
    print('%-60s dur %8.1f us  gap before %8.1f us  stream %s' % (r['Kernel_Name'][:60], (e - b) / 1e3, gap, r.get('Stream_Id', r.get('Queue_Id', '?'))))
    prev_end = e
PY
done
