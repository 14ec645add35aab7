mkdir -p gpurun_out/r04d
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_trunk.py -x -q > gpurun_out/r04d/tests.log 2>&1; echo tests rc=$?; tail -2 gpurun_out/r04d/tests.log
python tools/tune_conv.py --planes 3 --n 2000 --shapes 0 --cfgs 38,40,41 > gpurun_out/r04d/tune_l1.log 2>&1
python tools/tune_conv.py --planes 3 --n 2000 --shapes 0 --cfgs 38,40,41 --noresid >> gpurun_out/r04d/tune_l1.log 2>&1
cat gpurun_out/r04d/tune_l1.log
bash tools/ab_mode.sh 1 1025 > gpurun_out/r04d/ab_rows.log 2>&1; cat gpurun_out/r04d/ab_rows.log
