mkdir -p gpurun_out/r04n
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04n/gputests.log 2>&1; echo tests rc=$?; tail -3 gpurun_out/r04n/gputests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/collect_r04.sh r04y 2>&1 | tail -30
