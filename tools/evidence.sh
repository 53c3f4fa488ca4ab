#!/bin/bash
# The round's evidence on the GPU box, in parts (gpurun calls are limited to 20 minutes): tools/evidence.sh <part> [tag]   (tag: r04, r05 ...)
# (the one evidence script; earlier generations are parked in tools/attic/)
#   a: GPU tests + smoke + C2 (kernel trace, PMC passes, then the default bench line that quotes them)
#   b: C3 (bench, trace, PMC) + C4 bench + 2-rank rehearsal
#   c: host-pointer rates (read and write path), C5 (bench, trace at 12 500 entries, PMC), encode rate per class at levels 1 and 3
#   d: fuzz (damaged frames, LZ4 + Zstandard)
#   e: the LZ4 two-stage A/B of round 4 (one-kernel path / stage 2 over the output slot / stage 2 with the LDS window), stream rates
#   f: soak of the streaming write / read pair (600 random sizes, chunkings and windows)
#   g: scheduling and large entries (round 4): work lists largest first A/B (C4, C2, C3), ragged encode batch A/B, one 256 MiB entry
#   h: round 5: one large reference-made frame through the host read path and through the streaming reader
# Everything lands under gpurun_out/<tag>/; the PMC summaries are also copied to profiles/<tag>/ ON THE BOX so that the bench lines quote them.
part=${1:-a}; tag=${2:-r04}
out=gpurun_out/$tag
mkdir -p $out profiles/$tag
root=$PWD
trace() { name=$1; shift; (cd /tmp && export TMPDIR=/tmp && timeout -k 10 900 rocprofv3 --kernel-trace --stats -d $root/$out/trace_$name -o t --output-format csv -- python3 $root/bench.py "$@" --no-cpu > $root/$out/trace_${name}_bench.json 2> $root/$out/trace_$name.err; echo "trace $name rc=$?"; f=$(find $root/$out/trace_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $root/$out/${tag}_${name}_kernel_stats.csv); }
pmc2() { name=$1; wl=$2; n=$3; shift 3; (cd /tmp && export TMPDIR=/tmp && for grp in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM"; do g=$(echo $grp | cut -d' ' -f1); rm -rf $root/$out/pmc_$name/$g; timeout -k 10 600 rocprofv3 --pmc $grp -d $root/$out/pmc_$name/$g -o p --output-format csv -- python3 $root/bench.py "$@" --steps 2 --warmup 1 --no-cpu > $root/$out/pmc_${name}_$g.log 2>&1; echo "pmc $name $g rc=$?"; done); python tools/pmc_summary.py $out/pmc_$name --json $out/pmc_$wl.json --entries $n --workload $wl > $out/pmc_$wl.txt; cp $out/pmc_$wl.json $out/pmc_$wl.txt profiles/$tag/; }
line() { python3 - <<PY
import json
try:
    d=json.loads(open("$out/${tag}_$1_bench.json").read().strip().splitlines()[-1]); r=d["roofline"]; c=d.get("cpu_baseline") or {}
    print("$1", round(d["value"],1), d["unit"], round(d["ms_per_step"],2), "ms; frac", round(r["frac"],4), "of copy ceiling", r.get("frac_of_copy_ceiling") and round(r["frac_of_copy_ceiling"],4), "kernel_ms", round(r["kernel_ms"],3), r.get("stage_ms"), "cpu", c.get("value") and round(c["value"],2), "1T", (c.get("one_thread") or {}).get("value"), "traffic", r.get("traffic"), d["parity"])
except Exception as e: print("$1: no line", e)
PY
}
case $part in
a)
  timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $out/${tag}_pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $out/${tag}_pytest_gpu.log; tail -3 $out/${tag}_pytest_gpu.log
  timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/${tag}_smoke.log 2>&1; echo "smoke rc=$?"
  trace c2_lz4 --steps 10 --warmup 3
  pmc2 c2 c2_lz4_64k 100000
  timeout -k 10 900 python bench.py > $out/${tag}_c2_lz4_bench.json 2> $out/c2.err; echo "bench c2 rc=$?"; line c2_lz4 ;;
b)
  trace c3_zstd --workload c3_zstd_256k --steps 3 --warmup 1
  pmc2 c3 c3_zstd_256k 100000 --workload c3_zstd_256k
  timeout -k 10 900 python bench.py --workload c3_zstd_256k --steps 3 --warmup 1 > $out/${tag}_c3_zstd_bench.json 2> $out/c3.err; echo "bench c3 rc=$?"; line c3_zstd
  trace c4_mixed --workload c4_mixed --steps 3 --warmup 1
  pmc2 c4 c4_mixed 125000 --workload c4_mixed
  timeout -k 10 900 python bench.py --workload c4_mixed --steps 3 --warmup 1 > $out/${tag}_c4_mixed_bench.json 2> $out/c4.err; echo "bench c4 rc=$?"; line c4_mixed
  timeout -k 10 600 python bench.py --gpus 4 --workload c4_mixed --entries 40000 --steps 3 --warmup 1 --no-cpu > $out/${tag}_c4_4rank_40000_one_card_rehearsal.json 2> $out/strong.err; echo "4-rank rc=$?"; cut -c1-300 $out/${tag}_c4_4rank_40000_one_card_rehearsal.json
  timeout -k 10 600 python bench.py --gpus 2 --workload c5_zstd1_1m --entries 600 --steps 2 --warmup 1 --no-cpu > $out/${tag}_c5_2rank_one_card_rehearsal.json 2> $out/c5x2.err; echo "c5 2-rank rc=$?" ;;
b4)   # (only the C4 part of b)
  pmc2 c4 c4_mixed 125000 --workload c4_mixed
  timeout -k 10 900 python bench.py --workload c4_mixed --steps 3 --warmup 1 > $out/${tag}_c4_mixed_bench.json 2> $out/c4.err; echo "bench c4 rc=$?"; line c4_mixed ;;
c)
  { python3 tools/host_rate.py 20000 2 2>&1 | tail -2; python3 tools/host_rate.py 60000 2 2>&1 | tail -1; python3 tools/host_rate.py 8000 1 2>&1 | tail -1; } | tee $out/${tag}_host_rate.txt
  { python3 tools/host_write_rate.py 4000 1048576 1 1 2>&1 | grep -v amdgpu.ids | tail -3; python3 tools/host_write_rate.py 40000 65536 2 0 2>&1 | grep -v amdgpu.ids | tail -3; } | tee $out/${tag}_host_write_rate.txt
  trace c5_zstd1 --workload c5_zstd1_1m --steps 2 --warmup 1
  pmc2 c5 c5_zstd1_1m 12500 --workload c5_zstd1_1m
  timeout -k 10 1100 python bench.py --workload c5_zstd1_1m --steps 3 --warmup 1 > $out/${tag}_c5_zstd1_bench.json 2> $out/c5.err; echo "bench c5 rc=$?"; line c5_zstd1
  { timeout -k 10 300 python3 tools/enc_bench.py 4000 1048576 1 2>&1 | grep -E "^(text|records|random|runs) "; timeout -k 10 300 python3 tools/enc_bench.py 2000 1048576 3 2>&1 | grep -E "^(text|records|random|runs) "; } | tee $out/${tag}_enc_classes_levels.txt ;;
d)
  timeout -k 10 1100 python3 tools/fuzz_gpu.py 400 11 all > $out/${tag}_fuzz_all.log 2>&1; echo "fuzz rc=$?"; tail -8 $out/${tag}_fuzz_all.log ;;
d2)   # the same frames with every batch's work lists ordered (the default orders batches of >= 8192 entries only)
  ZPK_FUZZ_ORDER_MIN=1 timeout -k 10 1100 python3 tools/fuzz_gpu.py 400 13 all > $out/${tag}_fuzz_ordered.log 2>&1; echo "fuzz rc=$?"; tail -8 $out/${tag}_fuzz_ordered.log ;;
e)
  echo "part e (the LZ4 two-stage A/B of round 4) went with that path: tools/attic/r4_ab.sh" ;;
g)
  tools/attic/r4_order.sh 2>&1 | tee $out/${tag}_order_ab.txt
  timeout -k 10 600 python3 tools/enc_ragged.py 30000 1 2>&1 | grep -v amdgpu.ids | tee $out/${tag}_enc_ragged_order.txt
  timeout -k 10 400 python3 tools/big_entry_rate.py 256 16 2>&1 | grep -v amdgpu.ids | tee $out/${tag}_big_entry_rate.txt ;;
h)   # round 5: ONE large frame of the reference writer through the host read path (block-parallel) and through zpack_read_file_stream (bounded steps)
  timeout -k 10 300 python3 tools/big_frame_rate.py 256 16 lz4 2>&1 | grep -v amdgpu.ids | tee $out/${tag}_big_lz4_frame_rate.txt
  timeout -k 10 300 python3 tools/big_frame_rate.py 256 8 zstd 3 2>&1 | grep -v amdgpu.ids | tee $out/${tag}_big_zstd_frame_rate.txt
  timeout -k 10 300 python3 tools/big_frame_rate.py 256 8 zstd 1 2>&1 | grep -v amdgpu.ids | tee -a $out/${tag}_big_zstd_frame_rate.txt
  timeout -k 10 600 python3 tools/stream_rate.py lz4_0_64m_text,lz4_0_64m_records,lz4_0_512m_text,zstd_3_64m_text,zstd_3_64m_records,zstd_3_512m_text 2>&1 | grep -v amdgpu.ids | tee $out/${tag}_stream_read_rate.txt
  timeout -k 10 400 python3 tools/big_entry_rate.py 256 16 2>&1 | grep -v amdgpu.ids | tee $out/${tag}_big_entry_rate.txt
  timeout -k 10 300 python3 tools/mid_entry_rate.py 2>&1 | grep -v amdgpu.ids | tee $out/${tag}_mid_entry_rate.txt ;;
f)
  # soak of the streaming write / read pair: random entry sizes, chunkings and windows (the suite runs 30 of these; here 600 more, other seed)
  ZPK_STREAM_FUZZ_ITERS=600 ZPK_STREAM_FUZZ_SEED=7 timeout -k 10 1000 python -m pytest tests/test_gpu_zpack_api.py -x -q -m gpu -k random_sizes_and_windows > $out/${tag}_stream_soak.log 2>&1; echo "stream soak rc=$?" | tee -a $out/${tag}_stream_soak.log; tail -4 $out/${tag}_stream_soak.log
  ZPK_BIG_FUZZ_ITERS=250 timeout -k 10 1000 python -m pytest tests/test_gpu_big_entries.py -x -q -m gpu -k damage > $out/${tag}_big_entry_damage_soak.log 2>&1; echo "big-entry damage soak (3 x 250 flipped bytes: frame-parallel verdict == one-wave verdict) rc=$?" | tee -a $out/${tag}_big_entry_damage_soak.log; tail -3 $out/${tag}_big_entry_damage_soak.log ;;
esac
