O=$GRAFT_REPO_ROOT/gpurun_out/r05ax; mkdir -p $O
for i in 1 2 3; do
timeout -k 10 400 python bench.py --no-cpu-baseline --no-api-leg > $O/bench_cfg3_$i.json 2>/dev/null
python -c "
import json; d=json.load(open('$O/bench_cfg3_$i.json')); print('cfg3', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['timing']['python_gc'][:8], d['timing']['longest_step_ms'])"
done
WSI_BENCH_GC=on timeout -k 10 400 python bench.py --no-cpu-baseline --no-api-leg > $O/bench_cfg3_gcon.json 2>/dev/null
python -c "
import json; d=json.load(open('$O/bench_cfg3_gcon.json')); print('cfg3 gc on', d['value'], d['ms_per_step'], d['timing']['longest_step_ms'])"
timeout -k 10 400 python bench.py --workload cfg4 --no-cpu-baseline --no-api-leg > $O/bench_cfg4.json 2>/dev/null
python -c "
import json; d=json.load(open('$O/bench_cfg4.json')); print('cfg4', d['value'], d['ms_per_step'], d['timing']['longest_step_ms'])"
