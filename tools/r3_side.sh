#!/bin/bash
# A/B: LZ4 kernels beside the Zstandard stages (side stream) vs one stream, on the mixed workload, pure Zstandard and the headline
mkdir -p gpurun_out/r3d
tools/abl_run.sh --workload c4_mixed --steps 4 --warmup 2 --no-cpu; cp gpurun_out/abl_run.txt gpurun_out/r3d/abl_c4.txt
tools/abl_run.sh --workload c3_zstd_256k --steps 3 --warmup 2 --no-cpu; cp gpurun_out/abl_run.txt gpurun_out/r3d/abl_c3.txt
tools/abl_run.sh --steps 8 --warmup 3 --no-cpu; cp gpurun_out/abl_run.txt gpurun_out/r3d/abl_c2.txt
