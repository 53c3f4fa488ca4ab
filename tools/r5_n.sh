#!/bin/bash
# round 5, step N: large single LZ4 frames block-parallel (lz4_pj.h): tests, then the rate of one 256 MiB frame
out=gpurun_out/r05n; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_big_entries.py -m gpu -x -q > $out/pytest.log 2>&1; rc=$?
tail -6 $out/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python3 tools/big_frame_rate.py 256 16 2>&1 | grep -v amdgpu.ids | tee $out/big_frame_rate.txt
