#!/bin/bash
# A/B of codec builds on one box: the LZ4 tests with the default build, then every zpack_amd/abl_*.so on text / records / the mix
mkdir -p gpurun_out/r3b
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "${1:-lz4 or LZ4 or corrupt or foreign or status or device_batch or reference}" > gpurun_out/r3b/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3b/pytest.log
tools/abl_run.sh --entries 30000 --steps 3 --warmup 1 --no-cpu --mix 0
tools/abl_run.sh --entries 30000 --steps 3 --warmup 1 --no-cpu --mix 1
tools/abl_run.sh --entries 30000 --steps 3 --warmup 1 --no-cpu --mix 3
tools/abl_run.sh --steps 5 --warmup 2 --no-cpu
