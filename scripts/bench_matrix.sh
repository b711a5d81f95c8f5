#!/bin/bash
# Bench variants on one GPU box: pipelined vs serial loop, forced RCCL path, shard-sized corpora.
mkdir -p gpurun_out; : > gpurun_out/matrix.log
run() { # label, env, args
  echo "=== $1" >> gpurun_out/matrix.log
  env $2 timeout -k 10 300 python bench.py $3 --cpu-queries 0 --latency-queries 0 2>gpurun_out/matrix.err | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('qps', d['value'], 'ms/step', d['ms_per_step'], 'scan_ms', d['roofline']['mean_kernel_ms'], 'frac', d['roofline']['frac'])" >> gpurun_out/matrix.log 2>&1 || tail -5 gpurun_out/matrix.err >> gpurun_out/matrix.log
}
run "1M pipelined"        "X=1" "--steps 1000"
run "1M serial"           "DEWI_BENCH_SERIAL=1" "--steps 1000"
run "1M forced-dist"      "DEWI_BENCH_FORCE_DIST=1" "--steps 1000"
run "500K pipelined"      "X=1" "--docs 500000 --steps 2000"
run "500K forced-dist"    "DEWI_BENCH_FORCE_DIST=1" "--docs 500000 --steps 2000"
run "250K pipelined"      "X=1" "--docs 250000 --steps 3000"
run "250K forced-dist"    "DEWI_BENCH_FORCE_DIST=1" "--docs 250000 --steps 3000"
run "125K pipelined"      "X=1" "--docs 125000 --steps 4000"
run "125K serial"         "DEWI_BENCH_SERIAL=1" "--docs 125000 --steps 4000"
run "125K forced-dist"    "DEWI_BENCH_FORCE_DIST=1" "--docs 125000 --steps 4000"
cat gpurun_out/matrix.log
