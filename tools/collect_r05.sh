#!/bin/bash
# Round-5 evidence run on the MI355X box (one gpurun call): bench lines of every workload / mode, the rocprofv3 kernel-trace
# summary of the default command, the PMC traffic passes at the bench's own batch, per-launch times, trunk counters.
# Outputs under gpurun_out/<tag>/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r05z}
mkdir -p $O
python3 bench.py > $O/bench_cfg3_mx.json 2> $O/bench_cfg3_mx.err && echo "bench default done" &&
python3 bench.py --workload cfg2 --no-cpu-baseline --no-parity-leg > $O/bench_cfg2_mx.json 2> $O/bench_cfg2_mx.err &&
python3 bench.py --workload cfg4 --no-cpu-baseline > $O/bench_cfg4_mx.json 2> $O/bench_cfg4_mx.err &&
python3 bench.py --workload cfg5 --no-cpu-baseline > $O/bench_cfg5_mx.json 2> $O/bench_cfg5_mx.err &&
python3 bench.py --workload seg > $O/bench_seg_parity.json 2> $O/bench_seg_parity.err &&
python3 bench.py --mode parity --no-cpu-baseline > $O/bench_cfg3_parity.json 2> $O/bench_cfg3_parity.err &&
python3 bench.py --mode speed --no-cpu-baseline --no-parity-leg > $O/bench_cfg3_speed.json 2> $O/bench_cfg3_speed.err && echo "bench lines done" &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity-leg --no-api-leg > $O/bench_cfg3_mx_under_rocprof.json 2> $O/rocprof.err && echo "rocprof done" &&
bash tools/collect_traffic.sh r05_mx --mode mx --no-parity-leg --no-bf16-leg --no-api-leg > $O/traffic.log 2>&1 && cp gpurun_out/traffic_r05_mx.json $O/ && echo "traffic done" &&
python3 tools/launch_times.py --planes 3 --n 2000 > $O/launch_times_mx.txt 2>&1 && echo "launch times done" &&
bash tools/pmc_trunk.sh ${1:-r05z}_trunk --planes 3 --n 2000 && cp gpurun_out/pmc_${1:-r05z}_trunk/summary.txt $O/trunk_kernels_counters.txt && echo "counters done"
find $O/rocprof -name "*kernel_stats.csv" -exec cp {} $O/bench_cfg3_mx_kernel_stats.csv \;
python3 tools/rocprof_solo_stats.py $O/rocprof > $O/bench_cfg3_mx_kernel_solo_stats.csv 2>&1
rm -rf $O/rocprof/*/*kernel_trace.csv
ls $O
python3 -m pytest tests/test_gpu_margin.py -s -q > $O/margin.txt 2>&1; python3 tools/margin_json.py $O/margin.txt $O/margin_families.json "round-5 kernels: parity = fp16 pair" || true
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/calib -- tools/probes/fetch_calib96 1024 > $O/fetch_calib.log 2>&1; python3 - $O <<'P'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(float)
for f in glob.glob(sys.argv[1] + '/calib/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == 'FETCH_SIZE': acc[r['Kernel_Name'][:40]] += float(r['Counter_Value'])
open(sys.argv[1] + '/fetch_calib96.txt', 'w').write(open(sys.argv[1] + '/fetch_calib.log').read() + '\n'.join('%s FETCH_SIZE x 1024 = %d bytes tallied' % (k, v * 1024) for k, v in sorted(acc.items())) + '\n')
P
cat $O/fetch_calib96.txt; rm -rf $O/calib
