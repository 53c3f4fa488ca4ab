#!/bin/bash
# Round-end evidence on the GPU box: tests, smoke, the default bench line, the rocprofv3 kernel-trace summary
# of the same command and the PMC passes.  Everything lands under gpurun_out/<tag>/; copy what should be
# judged into profiles/<round>/.
tag=${1:-r01}
out=gpurun_out/$tag
mkdir -p $out
root=$PWD
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; echo "smoke rc=$?"
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"
tail -c 1200 $out/bench_default.json
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $root/$out/trace -o t --output-format csv -- python3 $root/bench.py --no-cpu > $root/$out/trace_bench.json 2> $root/$out/trace.err; echo "trace rc=$?")
tools/pmc.sh $out/pmc --steps 2 --warmup 1 --no-cpu
python tools/pmc_summary.py $out/pmc --json $out/pmc_c2_lz4_64k.json --entries 100000 --workload c2_lz4_64k | sed -n 1,40p
