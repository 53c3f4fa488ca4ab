#!/bin/bash
# build ablation variants of the codec (developer): tools/abl.sh NAME "DEFINES" ...   -> gpurun_out/../zpack_amd/abl_NAME.so
cd /root/repo
while [ $# -ge 2 ]; do
  name=$1; defs=$2; shift 2
  flags=""; for d in $defs; do flags="$flags -D$d"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function $flags -o zpack_amd/abl_$name.so zpack_amd/csrc/zpk_codec.hip 2>&1 | grep -i "error" 
  echo built abl_$name "($defs)"
done
