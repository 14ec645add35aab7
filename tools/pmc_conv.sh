#!/bin/bash
# PMC counters for the conv tuning harness (counters in their own passes, no trace domains).
# usage: tools/pmc_conv.sh <tag> <tune_conv.py args...>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_$1; shift
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/a -- python3 tools/tune_conv.py --rounds 1 --iters 2 "$@" > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -- python3 tools/tune_conv.py --rounds 1 --iters 2 "$@" > $OUT/b.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/c -- python3 tools/tune_conv.py --rounds 1 --iters 2 "$@" > $OUT/c.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_VMEM TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $OUT/d -- python3 tools/tune_conv.py --rounds 1 --iters 2 "$@" > $OUT/d.log 2>&1
echo done
