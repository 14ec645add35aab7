cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05final
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity-leg --no-api-leg > $O/bench_cfg3_mx_under_rocprof.json 2> $O/rocprof.err && echo "rocprof done" &&
bash tools/collect_traffic.sh r05_mx --mode mx --no-parity-leg --no-bf16-leg --no-api-leg > $O/traffic.log 2>&1 && cp gpurun_out/traffic_r05_mx.json $O/ && echo "traffic done" &&
python3 tools/launch_times.py --planes 3 --n 2000 > $O/launch_times_mx.txt 2>&1 && echo "launch times done" &&
bash tools/pmc_trunk.sh r05final_trunk --planes 3 --n 2000 && cp gpurun_out/pmc_r05final_trunk/summary.txt $O/trunk_kernels_counters.txt && echo "counters done"
find $O/rocprof -name "*kernel_stats.csv" -exec cp {} $O/bench_cfg3_mx_kernel_stats.csv \;
python3 tools/rocprof_solo_stats.py $O/rocprof > $O/bench_cfg3_mx_kernel_solo_stats.csv 2>&1
rm -rf $O/rocprof/*/*kernel_trace.csv
python3 -m pytest tests/test_gpu_margin.py -s -q > $O/margin.txt 2>&1; python3 tools/margin_json.py $O/margin.txt $O/margin_families.json "round-5 kernels: parity = fp16 pair" || true
ls $O
