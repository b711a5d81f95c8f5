set -e
mkdir -p gpurun_out/r04_b1
timeout -k 10 300 python -m pytest tests/test_sharded_gloo.py -x -q -m gpu > gpurun_out/r04_b1/tests.log 2>&1 || { tail -30 gpurun_out/r04_b1/tests.log; exit 1; }
tail -2 gpurun_out/r04_b1/tests.log
Q="--cpu-queries 0 --latency-queries 0"
python3 bench.py --config c3 $Q > gpurun_out/r04_b1/bench_c3.json 2> gpurun_out/r04_b1/c3.err
python3 bench.py --batch 256 $Q > gpurun_out/r04_b1/bench_c2_batch256_shadow.json 2> gpurun_out/r04_b1/b256.err
python3 bench.py --batch 32 $Q > gpurun_out/r04_b1/bench_c2_batch32.json 2> gpurun_out/r04_b1/b32.err
python3 bench.py --batch 32 --shadow 1 $Q > gpurun_out/r04_b1/bench_c2_batch32_shadow.json 2> gpurun_out/r04_b1/b32s.err
python3 bench.py --steps 20 --warmup 5 $Q > gpurun_out/r04_b1/bench_c2_steps20.json 2> gpurun_out/r04_b1/c2.err
for f in gpurun_out/r04_b1/bench_*.json; do python3 -c "
import json,sys
r=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], r['value'], r['ms_per_step'], r['roofline'].get('frac'), r['roofline'].get('kernel'))"; done
bash scripts/bench_dims.sh gpurun_out/r04_dims_v2
