#!/bin/bash
# round 5: large single Zstandard frames block-parallel (zstd_pj.h): tests, then the rate of one 256 MiB frame
out=gpurun_out/r05z; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_big_entries.py -m gpu -x -q -k "zstd_frame or lz4_frame_is" > $out/pytest.log 2>&1; rc=$?
tail -12 $out/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python3 tools/big_frame_rate.py 256 8 zstd 3 2>&1 | grep -v amdgpu.ids | tee $out/big_zstd_frame_rate.txt
