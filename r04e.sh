mkdir -p gpurun_out/r04e
echo "--- slab3 with scheduling fences (current) vs head: 16x16 layer-1 maps (cfg4 shape) and the 4 trunk shapes"
python tools/tune_conv.py --planes 3 --n 32000 --scale 4 --shapes 0 --cfgs 31,38 > gpurun_out/r04e/tune_slab3_new.log 2>&1
python tools/tune_conv.py --planes 3 --n 2000 --cfgs 30,31 >> gpurun_out/r04e/tune_slab3_new.log 2>&1
cp wsi_segmentation_pipeline_amd/lib/libwsi_hip.so /tmp/lib_current.so; cp tools/ablibs/libA_head.so wsi_segmentation_pipeline_amd/lib/libwsi_hip.so
python tools/tune_conv.py --planes 3 --n 32000 --scale 4 --shapes 0 --cfgs 31,38 > gpurun_out/r04e/tune_slab3_head.log 2>&1
python tools/tune_conv.py --planes 3 --n 2000 --cfgs 30,31 >> gpurun_out/r04e/tune_slab3_head.log 2>&1
cp /tmp/lib_current.so wsi_segmentation_pipeline_amd/lib/libwsi_hip.so
echo NEW; grep -v amdgpu.ids gpurun_out/r04e/tune_slab3_new.log; echo HEAD; grep -v amdgpu.ids gpurun_out/r04e/tune_slab3_head.log
python bench.py --workload cfg4 --no-cpu-baseline > gpurun_out/r04e/bench_cfg4.json 2> gpurun_out/r04e/bench_cfg4.err; echo cfg4 rc=$?
python bench.py --workload seg --no-cpu-baseline > gpurun_out/r04e/bench_seg.json 2> gpurun_out/r04e/bench_seg.err; echo seg rc=$?
python -m pytest tests -m gpu -x -q > gpurun_out/r04e/gputests.log 2>&1; echo tests rc=$?; tail -3 gpurun_out/r04e/gputests.log
