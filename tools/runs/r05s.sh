mkdir -p gpurun_out/r05s
for b in 6200 8300 6200 8300 6200 8300; do timeout -k 10 300 python bench.py --batch $b --no-cpu-baseline --no-parity-leg --no-bf16-leg --no-api-leg 2>/dev/null > /tmp/l.json; python - $b <<'P'
import json, sys
d = json.load(open('/tmp/l.json'))
print('batch cap', sys.argv[1], round(d['value'], 1), d['ms_per_step'], {k: round(v['avg_ms'], 3) for k, v in d['kernels'].items()})
P
done > gpurun_out/r05s/batch_ab.txt 2>&1
cat gpurun_out/r05s/batch_ab.txt
