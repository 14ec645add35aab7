mkdir -p gpurun_out/r04v
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04v/gputests.log 2>&1; echo tests rc=$?; tail -3 gpurun_out/r04v/gputests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/r04v/bench_default.json 2> gpurun_out/r04v/bench_default.err; echo bench rc=$?
python bench.py --mode speed --no-cpu-baseline --no-parity-leg > gpurun_out/r04v/bench_speed.json 2>/dev/null; echo speed rc=$?
python bench.py --mode parity --no-cpu-baseline > gpurun_out/r04v/bench_parity.json 2>/dev/null; echo parity rc=$?
python - <<'P'
import json
for f in ('bench_default','bench_speed','bench_parity'):
    d=json.loads(open('gpurun_out/r04v/%s.json'%f).read().strip().split('\n')[-1])
    print(f, d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('achieved'), (d.get('roofline_bf16') or {}).get('achieved'), {k: round(v['avg_ms'], 3) for k, v in d['kernels'].items()})
P
