#!/bin/bash
# Evidence for the second half of round 1 (two-stage Zstandard decode, K7): everything under gpurun_out/<tag>/.
tag=${1:-r01b}
out=gpurun_out/$tag
mkdir -p $out
root=$PWD
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; echo "smoke rc=$?"
timeout 900 python bench.py > $out/c2_lz4_bench.json 2> $out/c2_lz4_bench.err; echo "bench c2 rc=$?"
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $root/$out/trace_c2 -o t --output-format csv -- python3 $root/bench.py --no-cpu > $root/$out/trace_c2_bench.json 2> $root/$out/trace_c2.err; echo "trace c2 rc=$?")
timeout 900 python bench.py --workload c3_zstd_256k --steps 3 --warmup 1 > $out/c3_zstd_bench.json 2> $out/c3_zstd_bench.err; echo "bench c3 rc=$?"
(cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $root/$out/trace_c3 -o t --output-format csv -- python3 $root/bench.py --workload c3_zstd_256k --steps 3 --warmup 1 --no-cpu > $root/$out/trace_c3_bench.json 2> $root/$out/trace_c3.err; echo "trace c3 rc=$?")
(cd /tmp && export TMPDIR=/tmp && for grp in FETCH_SIZE WRITE_SIZE; do timeout 600 rocprofv3 --pmc $grp -d $root/$out/pmc_c3/$grp -o p --output-format csv -- python3 $root/bench.py --workload c3_zstd_256k --steps 2 --warmup 1 --no-cpu > $root/$out/pmc_c3_$grp.log 2>&1; echo "pmc c3 $grp rc=$?"; done)
python tools/pmc_summary.py $out/pmc_c3 --json $out/pmc_c3_zstd_256k.json --entries 100000 --workload c3_zstd_256k | sed -n 1,30p
timeout 900 python bench.py --workload c4_mixed --steps 3 --warmup 1 > $out/c4_mixed_bench.json 2> $out/c4_mixed_bench.err; echo "bench c4 rc=$?"
timeout 600 python tools/enc_bench.py 4000 1048576 > $out/c5_encode_4000x1MiB.log 2>&1; echo "enc rc=$?"
tail -n 12 $out/c5_encode_4000x1MiB.log
for f in c2_lz4 c3_zstd c4_mixed; do python - <<PY
import json
d=json.loads(open("$out/${f}_bench.json").read().strip().splitlines()[-1])
print("$f", round(d["value"],1), d["unit"], round(d["ms_per_step"],2), "ms; roofline", round(d["roofline"]["frac"],4), d["roofline"]["kernel_ms"], d["roofline"].get("stage_ms"), "cpu", d["cpu_baseline"] and round(d["cpu_baseline"]["value"],1), d.get("decode_stats"), d["parity"])
PY
done
