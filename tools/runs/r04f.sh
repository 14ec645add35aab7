mkdir -p gpurun_out/r04f
cp wsi_segmentation_pipeline_amd/lib/libwsi_hip.so /tmp/lib_current.so
bash tools/ab_lib.sh tools/ablibs/libC_rows.so tools/ablibs/libD_s2pref.so --streams 1 > gpurun_out/r04f/ab_s2pref.log 2>&1
cp /tmp/lib_current.so wsi_segmentation_pipeline_amd/lib/libwsi_hip.so
cat gpurun_out/r04f/ab_s2pref.log
for s in 1 2 3; do python bench.py --streams $s --no-cpu-baseline --no-parity-leg --no-bf16-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('streams', $s, d['value'], d['ms_per_step'])"; done | tee gpurun_out/r04f/streams.log
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_trunk.py -x -q > gpurun_out/r04f/tests.log 2>&1; echo tests rc=$?; tail -2 gpurun_out/r04f/tests.log
