mkdir -p gpurun_out/r04g
cp wsi_segmentation_pipeline_amd/lib/libwsi_hip.so /tmp/lib_current.so
cp tools/ablibs/libS_study.so wsi_segmentation_pipeline_amd/lib/libwsi_hip.so
python tools/wide_stamps.py --n 2000 > gpurun_out/r04g/wide_stamps.txt 2>&1
python tools/wide_stamps.py --n 2000 --cfg 38 --shape 64,64,64 >> gpurun_out/r04g/wide_stamps.txt 2>&1
python tools/wide_stamps.py --n 2000 --cfg 40 --shape 64,64,64 >> gpurun_out/r04g/wide_stamps.txt 2>&1
cp /tmp/lib_current.so wsi_segmentation_pipeline_amd/lib/libwsi_hip.so
grep -v amdgpu.ids gpurun_out/r04g/wide_stamps.txt
bash tools/ab_lib.sh tools/ablibs/libD_s2pref.so tools/ablibs/libE_stem.so --streams 1 > gpurun_out/r04g/ab_stem.log 2>&1
cp /tmp/lib_current.so wsi_segmentation_pipeline_amd/lib/libwsi_hip.so
cat gpurun_out/r04g/ab_stem.log
python -m pytest tests/test_gpu_trunk.py -x -q > gpurun_out/r04g/tests.log 2>&1; echo tests rc=$?; tail -2 gpurun_out/r04g/tests.log
