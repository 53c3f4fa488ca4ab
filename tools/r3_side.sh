#!/bin/bash
# A/B: LZ4 kernels beside the Zstandard stages (side stream) vs one stream, on the mixed workload and on the headline
mkdir -p gpurun_out/r3d
tools/abl_run.sh --workload c4_mixed --steps 3 --warmup 1 --no-cpu; cp gpurun_out/abl_run.txt gpurun_out/r3d/abl_c4.txt
tools/abl_run.sh --workload c4_mixed --steps 3 --warmup 1 --no-cpu --entries 40000; cp gpurun_out/abl_run.txt gpurun_out/r3d/abl_c4_40k.txt
tools/abl_run.sh --steps 5 --warmup 2 --no-cpu; cp gpurun_out/abl_run.txt gpurun_out/r3d/abl_c2.txt
