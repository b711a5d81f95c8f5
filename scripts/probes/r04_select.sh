set -e
export TMPDIR=/tmp
O=gpurun_out/r04_select; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_hip_search.py tests/test_hip_index_api.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
Q="--cpu-queries 0 --latency-queries 200"
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o bench -- python3 bench.py --steps 400 --warmup 30 --cpu-queries 0 --latency-queries 0 > $O/under_trace.json 2> $O/trace.err
python3 scripts/step_overhead.py $O/trace | tee $O/step_overhead.txt
rm -rf $O/trace
python3 bench.py --steps 20 --warmup 5 $Q > $O/bench_c2_steps20.json 2> $O/err.txt
python3 -c "
import json
r=json.loads(open('$O/bench_c2_steps20.json').read().strip().splitlines()[-1]); print(r['value'], r['ms_per_step'], r['roofline']['mean_kernel_ms'], r.get('p50_latency_ms'), r.get('c1_api'))"
