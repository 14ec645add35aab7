O=$GRAFT_REPO_ROOT/gpurun_out/r05ay; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; tail -3 $O/gputests.log
timeout -k 10 600 python bench.py > $O/bench_cfg3.json 2> $O/bench_cfg3.err || tail -5 $O/bench_cfg3.err
python -c "
import json; d=json.load(open('$O/bench_cfg3.json')); a=d['api']; print('cfg3', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], a['value'], a['vs_bare_engine_one_slide_per_call'], d['cpu_baseline']['value'], d['contract'])"
timeout -k 10 400 python bench.py --workload cfg4 --no-cpu-baseline --no-api-leg > $O/bench_cfg4.json 2>/dev/null
python -c "
import json; d=json.load(open('$O/bench_cfg4.json')); print('cfg4', d['value'], d['ms_per_step'], {k: round(v['avg_ms'],3) for k,v in d['kernels'].items()})"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
