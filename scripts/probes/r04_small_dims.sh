set -e
O=gpurun_out/r04_small; mkdir -p $O
Q="--cpu-queries 0 --latency-queries 0"
run() { local name=$1; shift; python3 bench.py "$@" $Q > $O/bench_$name.json 2> $O/$name.err || { tail -5 $O/$name.err; exit 1; }
  python3 -c "
import json
r=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1]); print('$name', r['value'], r['ms_per_step'], r['roofline'].get('kernel'), r['roofline'].get('mean_kernel_ms'), r['roofline'].get('frac'))"; }
run dim256_batch32 --dim 256 --docs 3000000 --batch 32 --steps 300 --warmup 60
run dim512_batch32 --dim 512 --docs 1500000 --batch 32 --steps 300 --warmup 60
run dim256_batch4 --dim 256 --docs 3000000 --batch 4 --steps 300 --warmup 60
run dim256_batch8 --dim 256 --docs 3000000 --batch 8 --steps 300 --warmup 60
run c2_batch2 --batch 2 --steps 300 --warmup 60
run c2_batch3 --batch 3 --steps 300 --warmup 60
run c2_batch4 --batch 4 --steps 300 --warmup 60
run dim384_batch4 --dim 384 --docs 2000000 --batch 4 --steps 300 --warmup 60
