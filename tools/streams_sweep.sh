#!/bin/bash
# A/B of batches in flight (HIP streams) and batch size on the default workload; prints patches/s per setting.
mkdir -p gpurun_out/streams
for cfg in "1 6200" "2 6200" "1 12400" "2 12400" "2 4200"; do
  set -- $cfg
  python bench.py --streams $1 --batch $2 --no-cpu-baseline --no-bf16-leg 2>/dev/null > gpurun_out/streams/s$1_b$2.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/streams/s$1_b$2.json").read().strip().splitlines()[-1]); print("streams", $1, "batch", $2, d["value"], d["ms_per_step"])
PY
done
