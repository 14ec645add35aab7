O=$GRAFT_REPO_ROOT/gpurun_out/r05au; mkdir -p $O
for m in mx speed; do
timeout -k 10 300 python bench.py --workload seg --mode $m --no-cpu-baseline --no-api-leg > $O/bench_seg_$m.json 2> $O/bench_seg_$m.err || tail -5 $O/bench_seg_$m.err
python -c "
import json; d=json.load(open('$O/bench_seg_$m.json')); print('$m', d['value'], d['ms_per_step'], d['dtype'][:30])"
done
for w in cfg2 cfg4 cfg5; do
timeout -k 10 400 python bench.py --workload $w --no-cpu-baseline --no-api-leg > $O/bench_$w.json 2> $O/bench_$w.err || tail -5 $O/bench_$w.err
python -c "
import json; d=json.load(open('$O/bench_$w.json')); print('$w', d['value'], d['unit'], d['ms_per_step'])"
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o seg -- python3 $GRAFT_REPO_ROOT/bench.py --workload seg --no-cpu-baseline --no-api-leg --no-parity-leg --steps 3 --warmup 1 > $O/bench_seg_under_rocprof.json 2> $O/stats.err
cd $GRAFT_REPO_ROOT
ls $O/stats/ | head; find $O/stats -name "*kernel_trace.csv" -delete
head -12 $(find $O/stats -name "*kernel_stats.csv" | head -1)
