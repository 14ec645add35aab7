mkdir -p gpurun_out/r05ap
for b in 128 256 512; do
timeout -k 10 300 python bench.py --workload seg --seg-batch $b --seg-tiles 512 --no-cpu-baseline --no-api-leg > gpurun_out/r05ap/bench_seg_$b.json 2> gpurun_out/r05ap/bench_seg_$b.err || tail -3 gpurun_out/r05ap/bench_seg_$b.err
python -c "
import json; d=json.load(open('gpurun_out/r05ap/bench_seg_$b.json')); print($b, d['value'], d['ms_per_step'], d.get('contract'), {k: (round(v['avg_ms'],3), v['launches']) for k,v in d['kernels'].items()})"
done
