cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05l
mkdir -p $O
python3 bench.py --mode parity --no-cpu-baseline > $O/bench_cfg3_parity.json 2> $O/bench_cfg3_parity.err &&
python3 bench.py --mode speed --no-cpu-baseline --no-parity-leg > $O/bench_cfg3_speed.json 2> $O/bench_cfg3_speed.err && echo "bench lines done" &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity-leg --no-api-leg > $O/bench_cfg3_mx_under_rocprof.json 2> $O/rocprof.err && echo "rocprof done" &&
bash tools/collect_traffic.sh r05_mx --mode mx --no-parity-leg --no-bf16-leg --no-api-leg > $O/traffic.log 2>&1 && cp gpurun_out/traffic_r05_mx.json $O/ && echo "traffic done" &&
python3 tools/launch_times.py --planes 3 --n 2000 > $O/launch_times_mx.txt 2>&1 && echo "launch times done" &&
bash tools/pmc_trunk.sh r05l_trunk --planes 3 --n 2000 && cp gpurun_out/pmc_r05l_trunk/summary.txt $O/trunk_kernels_counters.txt && echo "counters done"
find $O/rocprof -name "*kernel_stats.csv" -exec cp {} $O/bench_cfg3_mx_kernel_stats.csv \;
python3 tools/rocprof_solo_stats.py $O/rocprof > $O/bench_cfg3_mx_kernel_solo_stats.csv 2>&1
rm -rf $O/rocprof/*/*kernel_trace.csv
ls $O
