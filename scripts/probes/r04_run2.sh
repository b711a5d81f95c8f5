set -e
export TMPDIR=/tmp
O=gpurun_out/r04_b2; mkdir -p $O
Q="--cpu-queries 0 --latency-queries 0"
python3 bench.py --config c3 $Q > $O/bench_c3.json 2> $O/c3.err
python3 bench.py --batch 256 $Q > $O/bench_c2_batch256_shadow.json 2> $O/b256.err
python3 bench.py --batch 32 $Q > $O/bench_c2_batch32.json 2> $O/b32.err
for f in $O/bench_*.json; do python3 -c "
import json,sys
r=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], r['value'], r['ms_per_step'], r['roofline'].get('frac'), r['roofline'].get('kernel'))"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c3 -o bench -- python3 bench.py --config c3 --steps 40 --warmup 5 $Q > $O/under_trace_c3.json 2> $O/trace_c3.err
find $O/trace_c3 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_c3.csv
head -12 $O/kernel_stats_c3.csv | cut -c1-200
rm -rf $O/trace_c3
