mkdir -p gpurun_out/r04s
python tools/tune_conv.py --planes 2 --n 2000 --shapes 1,2,3 --cfgs 30,60,83 > gpurun_out/r04s/tune_parity.log 2>&1
python tools/tune_conv.py --planes 2 --n 2000 --shapes 1,2,3 --cfgs 30,60,83 --noresid >> gpurun_out/r04s/tune_parity.log 2>&1
grep -v amdgpu.ids gpurun_out/r04s/tune_parity.log
