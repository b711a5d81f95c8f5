# the depth-split pass with a partial last chunk: 32-query batches at dim 384 / 640 (fp32, 3 GB), and dim 768 as before
set -e
O=gpurun_out/r04_partial; mkdir -p $O
Q="--cpu-queries 0 --latency-queries 0"
run() { local name=$1; shift; python3 bench.py "$@" $Q > $O/bench_$name.json 2> $O/$name.err || { tail -5 $O/$name.err; exit 1; }
  python3 -c "
import json
r=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1]); print('$name', r['value'], r['ms_per_step'], r['roofline'].get('kernel'), r['roofline'].get('mean_kernel_ms'), r['roofline'].get('frac'), r.get('parity'))"; }
run dim384_batch32 --dim 384 --docs 2000000 --batch 32 --steps 300 --warmup 60
run dim640_batch32 --dim 640 --docs 1200000 --batch 32 --steps 300 --warmup 60
run dim128_batch32 --dim 128 --docs 6000000 --batch 32 --steps 300 --warmup 60
run dim1280_batch32 --dim 1280 --docs 600000 --batch 32 --steps 300 --warmup 60
run dim2048_batch32 --dim 2048 --docs 375000 --batch 32 --steps 300 --warmup 60
run dim1536_batch32 --dim 1536 --docs 500000 --batch 32 --steps 300 --warmup 60
run c2_batch32 --batch 32 --steps 400 --warmup 100
