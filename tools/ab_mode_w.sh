#!/bin/bash
# tools/ab_mode_w.sh <workload> mode mode ...: tools/ab_mode.sh for another bench workload
W=$1; shift
for rep in 1 2; do
  for m in "$@"; do
    python3 bench.py --workload $W --s2 $m --no-cpu-baseline --no-parity-leg --no-bf16-leg 2>/dev/null > /tmp/ab_line.json
    python3 - $m <<'P'
import json, sys
d = json.load(open('/tmp/ab_line.json'))
print('mode', sys.argv[1], round(d['value'], 1), d['unit'], d['ms_per_step'], {k: round(v['avg_ms'], 3) for k, v in d.get('kernels', {}).items()})
P
  done
done
