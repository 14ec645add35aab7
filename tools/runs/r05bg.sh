O=$GRAFT_REPO_ROOT/gpurun_out/r05bg; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/a -- python3 tools/seg_once.py --reps 1 > $O/a.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $O/b -- python3 tools/seg_once.py --reps 1 > $O/b.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA --output-format csv -d $O/d -- python3 tools/seg_once.py --reps 1 > $O/d.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_INSTS_WAVE32_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_WAIT_INST_ANY --output-format csv -d $O/e -- python3 tools/seg_once.py --reps 1 > $O/e.log 2>&1
python3 tools/pmc_summary.py $(find $O -name "*counter_collection.csv") > $O/summary.txt 2>&1
find $O -name "*counter_collection.csv" -delete
grep -A40 "unet_tail2" $O/summary.txt | head -60
