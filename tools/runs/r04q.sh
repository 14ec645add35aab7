mkdir -p gpurun_out/r04q
cp wsi_segmentation_pipeline_amd/lib/libwsi_hip.so /tmp/lib_current.so
bash tools/ab_lib.sh tools/ablibs/libF_head.so tools/ablibs/libG_early0.so --streams 1 > gpurun_out/r04q/ab_early0.log 2>&1
cp /tmp/lib_current.so wsi_segmentation_pipeline_amd/lib/libwsi_hip.so
cat gpurun_out/r04q/ab_early0.log
