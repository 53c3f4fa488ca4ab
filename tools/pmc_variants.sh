#!/bin/bash
# developer aid: instruction counters of k_lz4_wave for a list of codec builds (ablation variants).
# usage: tools/pmc_variants.sh <outdir> <so> [<so>...]
out=$1; shift
root=$PWD
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for so in "$@"; do
    name=$(basename "$so" .so)
    export ZPACK_AMD_CODEC_SO="$root/$so"
    timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU \
        -d "$root/$out/$name" -o p --output-format csv -- python3 "$root/bench.py" --entries 20000 --steps 2 --warmup 1 --no-cpu --skip-hash > "$root/$out/$name.log" 2>&1
    echo "== $name rc=$?"
    python3 "$root/tools/pmc_summary.py" "$root/$out/$name" | awk '/^k_lz4_wave/{f=1;next} /^k_/{f=0} f' | tr -s ' ' | tr '\n' ';'
    echo
done
