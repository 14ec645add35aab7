mkdir -p gpurun_out/r04c
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_trunk.py -x -q > gpurun_out/r04c/tests.log 2>&1; echo tests rc=$?; tail -2 gpurun_out/r04c/tests.log
cp wsi_segmentation_pipeline_amd/lib/libwsi_hip.so /tmp/lib_current.so
bash tools/ab_lib.sh tools/ablibs/libA_head.so tools/ablibs/libB_asmdma.so --streams 1 > gpurun_out/r04c/ab_asmdma.log 2>&1
cp /tmp/lib_current.so wsi_segmentation_pipeline_amd/lib/libwsi_hip.so
cat gpurun_out/r04c/ab_asmdma.log
python tools/tune_conv.py --planes 3 --n 2000 --cfgs 30,60 > gpurun_out/r04c/tune_wide.log 2>&1; cat gpurun_out/r04c/tune_wide.log
