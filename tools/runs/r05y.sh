mkdir -p gpurun_out/r05y
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05y/gputests.log 2>&1; tail -3 gpurun_out/r05y/gputests.log
timeout -k 10 400 python bench.py > gpurun_out/r05y/bench_default.json 2> gpurun_out/r05y/bench_default.err; python -c "
import json; d=json.load(open('gpurun_out/r05y/bench_default.json')); a=d['api']; print(d['value'], d['ms_per_step'], a['value'], a['vs_headline'], a['vs_bare_engine_one_slide_per_call'], a['bare_engine_one_slide_per_call']['value'], d['roofline']['achieved'], d['parity']['value'], d['contract'])"
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
