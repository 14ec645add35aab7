cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05final
mkdir -p $O
python3 bench.py > $O/bench_cfg3_mx.json 2> $O/bench_cfg3_mx.err && echo "bench default done" &&
python3 bench.py --workload cfg2 --no-cpu-baseline --no-parity-leg > $O/bench_cfg2_mx.json 2> $O/bench_cfg2_mx.err &&
python3 bench.py --workload cfg4 --no-cpu-baseline > $O/bench_cfg4_mx.json 2> $O/bench_cfg4_mx.err &&
python3 bench.py --workload cfg5 --no-cpu-baseline > $O/bench_cfg5_mx.json 2> $O/bench_cfg5_mx.err &&
python3 bench.py --workload seg > $O/bench_seg_parity.json 2> $O/bench_seg_parity.err &&
python3 bench.py --mode parity --no-cpu-baseline > $O/bench_cfg3_parity.json 2> $O/bench_cfg3_parity.err &&
python3 bench.py --mode speed --no-cpu-baseline --no-parity-leg > $O/bench_cfg3_speed.json 2> $O/bench_cfg3_speed.err && echo "bench lines done"
for f in $O/bench_*.json; do python3 -c "
import json,sys; d=json.load(open('$f')); print('$f', d['value'], d['unit'], d['ms_per_step'], (d.get('roofline') or {}).get('achieved'))"; done
