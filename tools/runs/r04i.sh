mkdir -p gpurun_out/r04i
run() { python bench.py "$@" --no-cpu-baseline --no-parity-leg --no-bf16-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
run --streams 1
run --streams 2 --no-stagger
run --streams 2
run --streams 2 --batch 3100
run --streams 2 --batch 2100
run --streams 3 --batch 3100
done | tee gpurun_out/r04i/stagger.log
python -m pytest tests/test_gpu_trunk.py tests/test_gpu_boundary.py tests/test_gpu_fullsize.py -x -q > gpurun_out/r04i/tests.log 2>&1; echo tests rc=$?; tail -2 gpurun_out/r04i/tests.log
