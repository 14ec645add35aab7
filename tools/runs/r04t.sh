mkdir -p gpurun_out/r04t
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04t/gputests.log 2>&1; echo tests rc=$?; tail -3 gpurun_out/r04t/gputests.log
python bench.py --mode parity --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('parity', d['value'], d['ms_per_step'], {k: round(v['avg_ms'], 3) for k, v in d['kernels'].items()})"
python bench.py --workload seg --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('seg', d['value'], d['ms_per_step'], 'mx leg', d['parity']['value'])"
