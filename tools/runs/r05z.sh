mkdir -p gpurun_out/r05z
run() { timeout -k 10 300 python bench.py --no-cpu-baseline --no-parity-leg --no-bf16-leg --no-api-leg "$@" 2>/dev/null > /tmp/l.json; python - "$@" <<'P'
import json, sys
d = json.load(open('/tmp/l.json'))
print(' '.join(sys.argv[1:]) or 'default', round(d['value'], 1), d['ms_per_step'], {k: round(v['avg_ms'], 3) for k, v in d['kernels'].items()})
P
}
for rep in 1 2; do run; run --streams 3; run --streams 1; run --stem 1,16; run --stem 1,64; done > gpurun_out/r05z/sweep.txt 2>&1
cat gpurun_out/r05z/sweep.txt
