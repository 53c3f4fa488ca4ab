#!/bin/bash
# round 5: streaming read in bounded block-parallel steps: the stream tests
out=gpurun_out/r05s; mkdir -p $out
timeout -k 10 1000 python -m pytest tests/test_gpu_codec.py tests/test_gpu_zpack_api.py -m gpu -x -q -k "stream" --durations=8 > $out/pytest.log 2>&1; rc=$?
tail -25 $out/pytest.log
exit $rc
