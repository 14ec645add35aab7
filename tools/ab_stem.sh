#!/bin/bash
# A/B of wsi_stem_set_mode settings on one box: tools/ab_stem.sh 1,32 4,32 ...  (each twice, interleaved; cfg3 default bench)
for rep in 1 2; do
  for m in "$@"; do
    python3 bench.py --stem $m --no-cpu-baseline --no-parity-leg --no-bf16-leg 2>/dev/null > /tmp/ab_line.json
    python3 - $m <<'P'
import json, sys
d = json.load(open('/tmp/ab_line.json'))
print('stem', sys.argv[1], round(d['value'], 1), d['ms_per_step'], {k: round(v['avg_ms'], 3) for k, v in d['kernels'].items()})
P
  done
done
