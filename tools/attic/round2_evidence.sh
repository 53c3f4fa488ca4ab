#!/bin/bash
# Round-2 evidence on the GPU box.  Everything lands under gpurun_out/r02/; copy what should be judged into profiles/r02/.
tag=${1:-r02}
out=gpurun_out/$tag
mkdir -p $out
root=$PWD
trace() { name=$1; shift; (cd /tmp && export TMPDIR=/tmp && timeout -k 10 900 rocprofv3 --kernel-trace --stats -d $root/$out/trace_$name -o t --output-format csv -- python3 $root/bench.py "$@" --no-cpu > $root/$out/trace_${name}_bench.json 2> $root/$out/trace_$name.err; echo "trace $name rc=$?"; f=$(find $root/$out/trace_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $root/$out/${tag}_${name}_kernel_stats.csv); }
pmc2() { name=$1; wl=$2; n=$3; shift 3; (cd /tmp && export TMPDIR=/tmp && for grp in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do g=$(echo $grp | cut -d' ' -f1); timeout -k 10 600 rocprofv3 --pmc $grp -d $root/$out/pmc_$name/$g -o p --output-format csv -- python3 $root/bench.py "$@" --steps 2 --warmup 1 --no-cpu > $root/$out/pmc_${name}_$g.log 2>&1; echo "pmc $name $g rc=$?"; done); python tools/pmc_summary.py $out/pmc_$name --json $out/pmc_$wl.json --entries $n --workload $wl > $out/pmc_$wl.txt; }
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $out/${tag}_pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $out/${tag}_pytest_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/${tag}_smoke.log 2>&1; echo "smoke rc=$?"
timeout -k 10 900 python bench.py > $out/${tag}_c2_lz4_bench.json 2> $out/c2.err; echo "bench c2 rc=$?"
trace c2_lz4 --steps 10 --warmup 3
pmc2 c2 c2_lz4_64k 100000
timeout -k 10 900 python bench.py --workload c3_zstd_256k --steps 3 --warmup 1 > $out/${tag}_c3_zstd_bench.json 2> $out/c3.err; echo "bench c3 rc=$?"
trace c3_zstd --workload c3_zstd_256k --steps 3 --warmup 1
pmc2 c3 c3_zstd_256k 100000 --workload c3_zstd_256k
timeout -k 10 900 python bench.py --workload c4_mixed --steps 3 --warmup 1 > $out/${tag}_c4_mixed_bench.json 2> $out/c4.err; echo "bench c4 rc=$?"
timeout -k 10 1100 python bench.py --workload c5_zstd1_1m --steps 3 --warmup 1 > $out/${tag}_c5_zstd1_bench.json 2> $out/c5.err; echo "bench c5 rc=$?"
trace c5_zstd1 --workload c5_zstd1_1m --entries 4000 --steps 2 --warmup 1
ZPK_BENCH_REHEARSAL=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 2 --workload c4_mixed --entries 30000 --scaling strong --steps 3 --warmup 1 > $out/${tag}_c4_strong_2rank_rehearsal.json 2> $out/strong.err; echo "strong2 rc=$?"
for f in c2_lz4 c3_zstd c4_mixed c5_zstd1; do python3 - <<PY
import json
try:
    d=json.loads(open("$out/${tag}_${f}_bench.json").read().strip().splitlines()[-1]); r=d["roofline"]; c=d.get("cpu_baseline") or {}
    print("$f", round(d["value"],1), d["unit"], round(d["ms_per_step"],2), "ms; frac", round(r["frac"],4), "of copy ceiling", r.get("frac_of_copy_ceiling") and round(r["frac_of_copy_ceiling"],4), "kernel_ms", round(r["kernel_ms"],3), r.get("stage_ms"), "cpu", c.get("value") and round(c["value"],2), "1T", (c.get("one_thread") or {}).get("value"), "traffic", r.get("traffic"), d["parity"])
except Exception as e: print("$f: no line", e)
PY
done
