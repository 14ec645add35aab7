#!/bin/bash
# PMC counters of every kernel of one trunk pass (tools/launch_times.py), counters in their own passes, no trace domains.
# usage: tools/pmc_trunk.sh <tag> [launch_times args]   ->  gpurun_out/pmc_<tag>/summary.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_$1; shift
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/a -- python3 tools/launch_times.py --reps 1 "$@" > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $OUT/b -- python3 tools/launch_times.py --reps 1 "$@" > $OUT/b.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA --output-format csv -d $OUT/d -- python3 tools/launch_times.py --reps 1 "$@" > $OUT/d.log 2>&1
python3 tools/pmc_summary.py $(find $OUT -name "*counter_collection.csv") > $OUT/summary.txt 2>&1
find $OUT -name "*counter_collection.csv" -delete
