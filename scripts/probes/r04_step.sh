set -e
export TMPDIR=/tmp
O=gpurun_out/r04_step; mkdir -p $O
Q="--cpu-queries 0 --latency-queries 0"
for K in 20 50 200 1000; do
  python3 bench.py --steps $K --warmup 5 $Q > $O/bench_c2_steps$K.json 2> $O/err_$K.txt
  python3 bench.py --steps $K --warmup 5 --condition-ms 0 $Q > $O/bench_c2_steps${K}_condition0.json 2> $O/err0_$K.txt
done
for f in $O/bench_c2_steps*.json; do python3 -c "
import json
r=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], r['value'], r['ms_per_step'], r['roofline']['mean_kernel_ms'], r['roofline']['timed_region'])"; done
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o bench -- python3 bench.py --steps 400 --warmup 30 $Q > $O/under_trace.json 2> $O/trace.err
python3 scripts/step_overhead.py $O/trace | tee $O/step_overhead.txt
rm -rf $O/trace
