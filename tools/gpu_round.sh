#!/bin/bash
# one gpurun call: the GPU test suite, then the default bench line; logs under gpurun_out/
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?
tail -5 gpurun_out/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
python bench.py --steps 10 --warmup 3 > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err || { tail -5 gpurun_out/bench_c2.err; exit 1; }
cat gpurun_out/bench_c2.json
