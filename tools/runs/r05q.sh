mkdir -p gpurun_out/r05q
timeout -k 10 400 python bench.py > gpurun_out/r05q/bench_default.json 2> gpurun_out/r05q/bench_default.err; python -c "
import json; d=json.load(open('gpurun_out/r05q/bench_default.json')); print(d['value'], d['api']['value'], d['api']['vs_headline'], d['api']['engine_defaults'], d['roofline']['achieved'], d['parity']['value'])"
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
