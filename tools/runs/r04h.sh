mkdir -p gpurun_out/r04h
bash tools/ab_mode.sh 1 16385 > gpurun_out/r04h/ab_lines128.log 2>&1; cat gpurun_out/r04h/ab_lines128.log
python tools/tune_conv.py --planes 3 --n 2000 --shapes 0 --cfgs 38,40 > gpurun_out/r04h/tune_l1.log 2>&1
python tools/tune_conv.py --planes 3 --n 2000 --shapes 0 --cfgs 38,40 --noresid >> gpurun_out/r04h/tune_l1.log 2>&1
grep -v amdgpu.ids gpurun_out/r04h/tune_l1.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r04h/calib -- tools/probes/fetch_calib96 1024 > gpurun_out/r04h/calib.log 2>&1
python - <<'P'
import csv, glob
for f in glob.glob('gpurun_out/r04h/calib/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == 'FETCH_SIZE': print(r['Kernel_Name'][:60], float(r['Counter_Value']) * 1024)
P
cat gpurun_out/r04h/calib.log | tail -2
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_trunk.py tests/test_gpu_unet.py -x -q > gpurun_out/r04h/tests.log 2>&1; echo tests rc=$?; tail -2 gpurun_out/r04h/tests.log
