set -e
cd $GRAFT_REPO_ROOT
bash tools/collect_traffic.sh r01f_mx --mode mx
cp gpurun_out/traffic_r01f_mx.json profiles/r01_traffic_mx.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r01f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r01f_bench_under_rocprof.json 2> gpurun_out/prof_r01f.err
timeout -k 10 400 python bench.py > gpurun_out/r01f_bench_mx.json 2> gpurun_out/r01f_bench_mx.err
timeout -k 10 400 python bench.py --mode parity --no-cpu-baseline > gpurun_out/r01f_bench_parity.json 2> /dev/null
timeout -k 10 400 python bench.py --mode speed --no-cpu-baseline > gpurun_out/r01f_bench_speed.json 2> /dev/null
timeout -k 10 400 python bench.py --streams 2 --no-cpu-baseline > gpurun_out/r01f_bench_mx_streams2.json 2> /dev/null
timeout -k 10 200 python tools/launch_times.py --planes 3 > gpurun_out/r01f_launch_times_mx.txt
timeout -k 10 200 python tools/launch_times.py --planes 2 > gpurun_out/r01f_launch_times_parity.txt
cp gpurun_out/prof_r01f/*/*_kernel_stats.csv gpurun_out/r01f_kernel_stats.csv
