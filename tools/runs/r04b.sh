mkdir -p gpurun_out/r04b
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_trunk.py tests/test_gpu_reference_pins.py -x -q -s > gpurun_out/r04b/tests.log 2>&1; echo tests rc=$?
python tools/tune_conv.py --planes 3 --n 2000 --shapes 0 --cfgs 38,40,41 > gpurun_out/r04b/tune_l1.log 2>&1
python tools/tune_conv.py --planes 3 --n 2000 --shapes 0 --cfgs 38,40,41 --noresid >> gpurun_out/r04b/tune_l1.log 2>&1
bash tools/ab_mode.sh 1 1025 > gpurun_out/r04b/ab_rows.log 2>&1
python bench.py > gpurun_out/r04b/bench_default.json 2> gpurun_out/r04b/bench_default.err; echo bench rc=$?
python bench.py --streams 1 --no-cpu-baseline --no-parity-leg --no-bf16-leg > gpurun_out/r04b/bench_s1.json 2>&1
python bench.py --workload seg > gpurun_out/r04b/bench_seg.json 2> gpurun_out/r04b/bench_seg.err; echo seg rc=$?
tail -3 gpurun_out/r04b/tests.log; cat gpurun_out/r04b/tune_l1.log gpurun_out/r04b/ab_rows.log
