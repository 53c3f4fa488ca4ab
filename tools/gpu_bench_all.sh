#!/bin/bash
# one gpurun call: bench lines of the round (driver-run default first), into gpurun_out/
mkdir -p gpurun_out
run() { name=$1; shift; timeout -k 10 900 python bench.py "$@" > gpurun_out/$name.json 2> gpurun_out/$name.err; echo "$name rc=$?"; python3 -c "
import json,sys
try:
    d=json.loads(open('gpurun_out/$name.json').readline())
    r=d['roofline']; c=d.get('cpu_baseline') or {}
    print('   value %.1f %s  ms/step %.2f  frac %.4f  of-copy-ceiling %s  kernel_ms %.3f stage %s  cpu %s (1T %s)  parity %s' % (d['value'], d['unit'], d['ms_per_step'], r['frac'], r.get('frac_of_copy_ceiling'), r['kernel_ms'], [round(x,2) for x in r['stage_ms']], c.get('value'), (c.get('one_thread') or {}).get('value'), d['parity']))
except Exception as e: print('   (no line)', e)
"; }
for a in "$@"; do
  case $a in
    c2) run bench_c2 --steps 10 --warmup 3;;
    c3) run bench_c3 --workload c3_zstd_256k --steps 5 --warmup 2;;
    c4) run bench_c4 --workload c4_mixed --steps 5 --warmup 2;;
    c5small) run bench_c5_small --workload c5_zstd1_1m --entries 1500 --steps 3 --warmup 1 --cpu-seconds 8;;
    c5) run bench_c5 --workload c5_zstd1_1m --steps 3 --warmup 1;;
    strong2) ZPK_BENCH_REHEARSAL=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 2 --workload c4_mixed --entries 20000 --scaling strong --steps 3 --warmup 1 > gpurun_out/bench_strong2.json 2> gpurun_out/bench_strong2.err; echo "strong2 rc=$?"; tail -c 600 gpurun_out/bench_strong2.json;;
  esac
done
