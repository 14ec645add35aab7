#!/bin/bash
# HBM traffic of the bench's kernels from PMC counters, per MI355X_MICROARCH.md: FETCH_SIZE and
# WRITE_SIZE in separate passes (no trace domains), FETCH doubled on gfx950 (128-B requests are
# tallied at 64 B for wide coalesced reads), WRITE as is.  Collected at the bench's OWN batch (r04; r01-r03 collected at
# batch 1000 and scaled): pass --batch to override; tools/traffic_json.py records the tiles per launch.  Writes gpurun_out/traffic_<tag>.json.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
OUT=gpurun_out/pmc_bench_$tag
mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-prof --streams 1 "$@" > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-prof --streams 1 "$@" > $OUT/write.log 2>&1
python3 tools/traffic_json.py $OUT gpurun_out/traffic_$tag.json "$@"
