set -e
O=gpurun_out/r04_shadow_dims; mkdir -p $O
Q="--cpu-queries 0 --latency-queries 0"
run() { local name=$1; shift; python3 bench.py "$@" $Q > $O/bench_$name.json 2> $O/$name.err || { tail -5 $O/$name.err; exit 1; }
  python3 -c "
import json
r=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1]); print('$name', r['value'], r['ms_per_step'], r['roofline'].get('kernel'), r['roofline'].get('mean_kernel_ms'), r['roofline'].get('frac'), r.get('parity'))"; }
run dim384_batch256_shadow --dim 384 --docs 2000000 --batch 256 --steps 200 --warmup 40
run dim384_batch32_shadow --dim 384 --docs 2000000 --batch 32 --shadow 1 --steps 200 --warmup 40
run dim1280_batch32_shadow --dim 1280 --docs 600000 --batch 32 --shadow 1 --steps 200 --warmup 40
