mkdir -p gpurun_out/r04r
run() { python bench.py "$@" --no-cpu-baseline --no-parity-leg --no-bf16-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'], {k: round(v['avg_ms'], 3) for k, v in d['kernels'].items()})"; }
for rep in 1 2; do
run --batch 6200
run --batch 8300
run --batch 12400
run --batch 4200
run --s2 257
run --s2 513
done | tee gpurun_out/r04r/sweep.log
