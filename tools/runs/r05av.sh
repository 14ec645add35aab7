O=$GRAFT_REPO_ROOT/gpurun_out/r05av; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for d in _old .; do
tag=$(echo $d | tr -d './_'); tag=${tag:-new}
(cd $GRAFT_REPO_ROOT/$d && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$tag -o c4 -- python3 bench.py --workload cfg4 --no-cpu-baseline --no-api-leg --no-prof --steps 3 --warmup 1 > $O/bench_$tag.json 2> $O/err_$tag.txt)
find $O/stats_$tag -name "*kernel_trace.csv" -delete
echo "== $tag"; python3 -c "
import json; d=json.load(open('$O/bench_$tag.json')); print(d['value'], d['ms_per_step'])"
python3 - $O/stats_$tag <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print('%-70s calls %5s total %9.3f ms avg %8.1f us' % (r['Name'][:70], r['Calls'], int(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3))
PY
done
