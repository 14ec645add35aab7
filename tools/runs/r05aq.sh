mkdir -p gpurun_out/r05aq
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05aq/gputests.log 2>&1; tail -3 gpurun_out/r05aq/gputests.log
timeout -k 10 500 python bench.py --workload seg > gpurun_out/r05aq/bench_seg.json 2> gpurun_out/r05aq/bench_seg.err || tail -5 gpurun_out/r05aq/bench_seg.err
python -c "
import json; d=json.load(open('gpurun_out/r05aq/bench_seg.json')); a=d['api']; print(d['value'], d['ms_per_step'], d['roofline']['achieved'], a['value'], a['ms_per_slide'], a['vs_bare_engine_one_slide_per_call'], a['generic_iterator_path']['value']); print(d['cpu_baseline']); print(d.get('contract'))"
