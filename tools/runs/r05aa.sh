mkdir -p gpurun_out/r05aa
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05aa/gputests.log 2>&1; tail -3 gpurun_out/r05aa/gputests.log
for w in cfg3 cfg4 seg; do timeout -k 10 400 python bench.py --workload $w --no-cpu-baseline --no-api-leg > gpurun_out/r05aa/bench_$w.json 2> gpurun_out/r05aa/bench_$w.err; python -c "
import json; d=json.load(open('gpurun_out/r05aa/bench_$w.json')); print('$w', d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('achieved'), {k: round(v['avg_ms'],3) for k,v in d['kernels'].items()})"; done
