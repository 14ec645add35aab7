#!/bin/bash
# A/B of two builds of the library on one box: tools/ab_lib.sh libA libB [bench args...]  (each twice, interleaved)
A=$1; B=$2; shift 2
L=wsi_segmentation_pipeline_amd/lib/libwsi_hip.so
for rep in 1 2; do
  for f in $A $B; do
    cp $f $L
    python3 bench.py "$@" --no-cpu-baseline --no-parity-leg --no-bf16-leg 2>/dev/null > /tmp/ab_line.json
    python3 - $f <<'P'
import json, sys
d = json.load(open('/tmp/ab_line.json'))
print(sys.argv[1].split('/')[-1], round(d['value'], 1), d['ms_per_step'], {k: round(v['avg_ms'], 3) for k, v in d['kernels'].items()})
P
  done
done
