set -e
python3 bench.py --batch 256 --steps 300 --warmup 60 --cpu-queries 64 > gpurun_out/r3_c2_b256.json 2> gpurun_out/r3_c2_b256.err || { tail -20 gpurun_out/r3_c2_b256.err; exit 1; }
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r3_c2_b256.json'))
print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['mean_kernel_ms'], d['roofline']['frac'], d['parity'], d['config'].get('bf16_shadow'))
PY
python3 bench.py --batch 64 --steps 300 --warmup 60 --cpu-queries 0 > gpurun_out/r3_c2_b64.json 2>/dev/null
python3 -c "
import json
d=json.load(open('gpurun_out/r3_c2_b64.json')); print('b64', d['value'], d['ms_per_step'], d['roofline']['mean_kernel_ms'])"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_trace_b256 -o bench -- python3 bench.py --batch 256 --steps 60 --warmup 10 --cpu-queries 0 > /dev/null 2> gpurun_out/r3_trace_b256.err
find gpurun_out/r3_trace_b256 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r3_kernel_stats_b256.csv
python3 - <<'PY'
import csv
for i,r in enumerate(csv.DictReader(open('gpurun_out/r3_kernel_stats_b256.csv'))):
    if i>=6: break
    print(r["Name"].split("(")[0][:60], r["Calls"], round(float(r["AverageNs"])/1e3,2), round(float(r["MinNs"])/1e3,2))
PY
