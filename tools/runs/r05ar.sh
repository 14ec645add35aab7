mkdir -p gpurun_out/r05ar
timeout -k 10 600 python -m pytest tests/test_gpu_unet.py tests/test_gpu_multirank.py -m gpu -x -q > gpurun_out/r05ar/gputests.log 2>&1; tail -2 gpurun_out/r05ar/gputests.log
timeout -k 10 500 python bench.py --workload seg --no-cpu-baseline > gpurun_out/r05ar/bench_seg.json 2> gpurun_out/r05ar/bench_seg.err || tail -5 gpurun_out/r05ar/bench_seg.err
python -c "
import json; d=json.load(open('gpurun_out/r05ar/bench_seg.json')); a=d['api']; print(d['value'], d['ms_per_step'], a['value'], a['ms_per_slide'], a['vs_bare_engine_one_slide_per_call'], a['generic_iterator_path']['value'], a['generic_iterator_path']['class_map_pixels_differing_from_fused_path'])"
