set -e
cd $GRAFT_REPO_ROOT
bash tools/collect_traffic.sh r01_mx --mode mx
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r01d -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r01d_bench_under_rocprof.json 2> gpurun_out/prof_r01d.err
timeout -k 10 400 python bench.py > gpurun_out/r01d_bench_mx.json 2> gpurun_out/r01d_bench_mx.err
timeout -k 10 400 python bench.py --mode parity --no-cpu-baseline > gpurun_out/r01d_bench_parity.json 2> /dev/null
timeout -k 10 400 python bench.py --mode speed --no-cpu-baseline > gpurun_out/r01d_bench_speed.json 2> /dev/null
timeout -k 10 400 python bench.py --streams 2 --no-cpu-baseline > gpurun_out/r01d_bench_mx_streams2.json 2> /dev/null
ls gpurun_out/prof_r01d/*/ | head
