O=$GRAFT_REPO_ROOT/gpurun_out/r05ae; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o seg -- python3 $GRAFT_REPO_ROOT/tools/seg_once.py --reps 3 > $O/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r05ae/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
st = [i for i, n in enumerate(names) if 'stem_pool' in n]
out = open('gpurun_out/r05ae/seg_batch.txt', 'w')
tot = 0
for r in rows[st[-1]:]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    out.write('%-72s grid %-9s wg %-4s %8.1f us\n' % (r['Kernel_Name'][:72], r['Grid_Size_X'], r['Workgroup_Size_X'], d))
out.write('sum %.1f us\n' % tot)
out.close()
PY
cat gpurun_out/r05ae/seg_batch.txt | tail -16
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/a -- python3 tools/seg_once.py --reps 1 > $O/a.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $O/b -- python3 tools/seg_once.py --reps 1 > $O/b.log 2>&1
python3 tools/pmc_summary.py $(find $O -name "*counter_collection.csv") > $O/summary.txt 2>&1
find $O -name "*counter_collection.csv" -delete
grep -B2 -A14 "unet_tail" $O/summary.txt | head -60
