#!/bin/bash
# HBM traffic counters (separate passes, as MI355X_MICROARCH.md prescribes): FETCH_SIZE / WRITE_SIZE in KiB units.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_$1; shift
mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 tools/tune_conv.py --rounds 1 --iters 2 "$@" > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 tools/tune_conv.py --rounds 1 --iters 2 "$@" > $OUT/write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/l2 -- python3 tools/tune_conv.py --rounds 1 --iters 2 "$@" > $OUT/l2.log 2>&1
echo done
