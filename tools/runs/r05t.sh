mkdir -p gpurun_out/r05t
timeout -k 10 700 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_multirank.py tests/test_gpu_margin.py tests/test_gpu_fullsize.py -x -q > gpurun_out/r05t/tests.log 2>&1; tail -3 gpurun_out/r05t/tests.log
for i in 1 2; do timeout -k 10 400 python bench.py --no-cpu-baseline --no-parity-leg --no-bf16-leg > gpurun_out/r05t/bench_default_$i.json 2> gpurun_out/r05t/bench_default.err; python -c "
import json; d=json.load(open('gpurun_out/r05t/bench_default_$i.json')); print(d['value'], d['api']['value'], d['api']['vs_headline'], d['api']['ms_per_slide'], d['ms_per_step'], d['api']['precision'])"; done
for b in 6200 8300 6200 8300; do timeout -k 10 300 python bench.py --batch $b --no-cpu-baseline --no-parity-leg --no-bf16-leg --no-api-leg 2>/dev/null > /tmp/l.json; python - $b <<'P'
import json, sys
d = json.load(open('/tmp/l.json'))
print('batch cap', sys.argv[1], round(d['value'], 1), d['ms_per_step'])
P
done > gpurun_out/r05t/batch_ab.txt 2>&1
cat gpurun_out/r05t/batch_ab.txt
