#!/bin/bash
# A/B of codec builds on one box: GPU tests (-k "$1", default all) with the default build, then every zpack_amd/abl_*.so on text and on the C2 mix (+ "$2" extra bench workloads)
mkdir -p gpurun_out/r3b
timeout -k 10 900 python -m pytest tests -m gpu -x -q ${1:+-k "$1"} > gpurun_out/r3b/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3b/pytest.log
tools/abl_run.sh --entries 30000 --steps 3 --warmup 1 --no-cpu --mix 0; cp gpurun_out/abl_run.txt gpurun_out/r3b/abl_text.txt
tools/abl_run.sh --steps 5 --warmup 2 --no-cpu; cp gpurun_out/abl_run.txt gpurun_out/r3b/abl_mix.txt
for w in $2; do tools/abl_run.sh --workload $w --steps 3 --warmup 1 --no-cpu --entries ${3:-20000}; cp gpurun_out/abl_run.txt gpurun_out/r3b/abl_$w.txt; done
