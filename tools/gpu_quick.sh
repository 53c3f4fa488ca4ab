#!/bin/bash
# quick GPU iteration: selected tests (-k "$1"), then optional bench args ("$2")
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "$1" > gpurun_out/pytest_quick.log 2>&1; rc=$?
tail -25 gpurun_out/pytest_quick.log
[ $rc -eq 0 ] || exit $rc
if [ -n "$2" ]; then timeout -k 10 600 python bench.py $2 > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err || { tail -5 gpurun_out/bench_quick.err; exit 1; }; cat gpurun_out/bench_quick.json; fi
