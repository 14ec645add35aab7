#!/bin/bash
# A/B of bench --batch caps on one box: tools/ab_batch.sh 6200 6656 ...  (each twice, interleaved; cfg3 default bench)
for rep in 1 2; do
  for m in "$@"; do
    python3 bench.py --batch $m --no-cpu-baseline --no-parity-leg --no-bf16-leg 2>/dev/null > /tmp/ab_line.json
    python3 - $m <<'P'
import json, sys
d = json.load(open('/tmp/ab_line.json'))
print('batch cap', sys.argv[1], round(d['value'], 1), d['ms_per_step'], {k: (round(v['avg_ms'], 3), v['launches']) for k, v in d['kernels'].items()})
P
  done
done
