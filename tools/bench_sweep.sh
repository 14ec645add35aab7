#!/bin/bash
# usage: tools/bench_sweep.sh tag "args1" "args2" ...   -> one summary line per arg set
tag=$1; shift
i=0
for a in "$@"; do
  i=$((i+1))
  timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline $a > gpurun_out/bench_${tag}_$i.log 2>&1
  python - "$a" gpurun_out/bench_${tag}_$i.log <<'PY'
import json, sys
ls=[x for x in open(sys.argv[2]) if x.startswith("{")]
if not ls:
    print(sys.argv[1], "FAILED"); print(open(sys.argv[2]).read()[-800:])
else:
    d=json.loads(ls[0])
    print("%-28s %8.0f p/s  s1 %.0f TF |" % (sys.argv[1], d["value"], d["roofline"]["achieved"] if d["roofline"] else 0), {k:(round(v["avg_ms"],3), round(v["share_of_step"],3)) for k,v in d["kernels"].items()})
PY
done
