mkdir -p gpurun_out/r05u
for i in 1 2; do timeout -k 10 400 python bench.py --no-cpu-baseline --no-parity-leg --no-bf16-leg > gpurun_out/r05u/bench_default_$i.json 2> gpurun_out/r05u/bench_default.err; python -c "
import json; d=json.load(open('gpurun_out/r05u/bench_default_$i.json')); a=d['api']; print(d['value'], d['ms_per_step'], a['value'], a['ms_per_slide'], a['vs_headline'], a['bare_engine_one_slide_per_call'], a['vs_bare_engine_one_slide_per_call'])"; done
