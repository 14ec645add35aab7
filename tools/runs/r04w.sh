mkdir -p gpurun_out/r04w
cp wsi_segmentation_pipeline_amd/lib/libwsi_hip.so /tmp/lib_current.so
bash tools/ab_lib.sh tools/ablibs/libH_head.so tools/ablibs/libI_s2nt4.so --streams 1 > gpurun_out/r04w/ab_s2nt4.log 2>&1
cp /tmp/lib_current.so wsi_segmentation_pipeline_amd/lib/libwsi_hip.so
cat gpurun_out/r04w/ab_s2nt4.log
python -m pytest tests/test_gpu_trunk.py tests/test_gpu_kernels.py -x -q 2>&1 | tail -2
