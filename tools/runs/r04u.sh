mkdir -p gpurun_out/r04u
python tools/tune_conv.py --planes 1 --n 2000 --shapes 1,2,3 --cfgs 30,60,70 > gpurun_out/r04u/tune_speed.log 2>&1
python tools/tune_conv.py --planes 1 --n 2000 --shapes 1,2,3 --cfgs 30,60,70 --noresid >> gpurun_out/r04u/tune_speed.log 2>&1
grep -v amdgpu.ids gpurun_out/r04u/tune_speed.log
