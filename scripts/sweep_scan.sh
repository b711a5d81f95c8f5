#!/bin/bash
# Launch-shape sweep of the scan kernel on the GPU box: blocks, rows per iteration, nontemporal loads.
# usage: bash scripts/sweep_scan.sh "<blocks> <rows_per_iter> <nt>" ...
mkdir -p gpurun_out
: > gpurun_out/sweep.log
for cfg in "$@"; do
  set -- $cfg
  echo "=== blocks=$1 R=$2 nt=$3" >> gpurun_out/sweep.log
  timeout -k 10 120 python bench.py --steps 400 --warmup 30 --cpu-queries 0 --latency-queries 0 --scan-blocks $1 --rows-per-iter $2 --nontemporal $3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('qps', d['value'], 'ms/step', d['ms_per_step'], 'scan_ms', d['roofline']['mean_kernel_ms'], 'frac', d['roofline']['frac'])" >> gpurun_out/sweep.log
done
cat gpurun_out/sweep.log
