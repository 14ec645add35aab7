O=$GRAFT_REPO_ROOT/gpurun_out/r05at; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_unet.py -m gpu -x -q > $O/gputests.log 2>&1; tail -2 $O/gputests.log
timeout -k 10 500 python bench.py --workload seg --no-cpu-baseline > $O/bench_seg.json 2> $O/bench_seg.err || tail -5 $O/bench_seg.err
python -c "
import json; d=json.load(open('$O/bench_seg.json')); a=d['api']; print(d['value'], d['ms_per_step'], a['value'], a['ms_per_slide'], a['vs_bare_engine_one_slide_per_call'], a['bare_pipeline_one_slide_per_call']['value'], a['vs_bare_pipeline_one_slide_per_call'], a['generic_iterator_path']['value'])"
